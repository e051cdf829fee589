"""Throughput of independent pairs when P of them are in flight at once (the reference's multi-pair loop is embarrassingly
parallel: run_hierarchical_optimizer3d_multipair.py:403-432): P host threads, each with its own optimizer and HIP stream,
run `steps` optimize() calls on pairs of their own.  One iteration launch fills every CU with a 1024-thread workgroup, so a
second stream's launch moves into the CUs as the first one's workgroups retire: the drain of one launch and the gap to the
next are filled by the other pair.  usage: pairs_in_flight.py [size] [steps] [iterations] [slavcheva | hier-full | hier-tik]"""
import gc
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
kind = sys.argv[4] if len(sys.argv) > 4 else "slavcheva"
canonical, live0 = sphere_pair(n, 3, "cuda")


def make_optimizer():
    if kind == "slavcheva":
        return lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                        smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                        maximum_warp_length_lower_threshold=0.0, max_iterations=iters,
                                        min_iterations=iters, check_interval=iters)
    return lsf.HierarchicalOptimizer3d(tikhonov_term_enabled=True, tikhonov_strength=0.05,
                                       gradient_kernel_enabled=kind == "hier-full",
                                       kernel=lsf.generate_1d_sobolev_kernel(7, 0.1) if kind == "hier-full" else None,
                                       maximum_chunk_size=8, rate=0.1, maximum_iteration_count=iters,
                                       maximum_warp_update_threshold=0.0)


def call(opt, live):
    if kind == "slavcheva":
        live.copy_(live0)
        opt.optimize(live, canonical)
        return live
    return opt.optimize(canonical, live0)


def worker(stream, opt, count, barrier, out, k):
    with torch.cuda.stream(stream):
        live = torch.empty_like(live0)
        for _ in range(2):
            call(opt, live)
        stream.synchronize()
        barrier.wait()
        for _ in range(count):
            result = call(opt, live)
        stream.synchronize()
        out[k] = float(result.double().sum().item())


for p in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(p)]
    barrier = threading.Barrier(p + 1)
    out = [None] * p
    # every optimizer makes its first calls ALONE (HIP graphs of the hierarchical optimizer's launch-bound levels are
    # captured then: HIP refuses other threads' calls while a capture is in progress), as experiment/multipair.py does
    optimizers = [make_optimizer() for _ in range(p)]
    for opt in optimizers:
        call(opt, torch.empty_like(live0))
        if hasattr(opt.engine, "allow_graph_capture"):
            opt.engine.allow_graph_capture = False
    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(streams[k], optimizers[k], steps, barrier, out, k)) for k in range(p)]
    for t in threads:
        t.start()
    barrier.wait()  # every lane has warmed up
    gc.collect()
    gc.freeze()
    gc.disable()  # a full collection walks every object torch has made: ~40 ms, i.e. one 2 ms step in 25 (bench.py)
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    gc.enable()
    print("%s, %d pair(s) in flight: %.3f ms per pair (checksums %s)" % (
        kind, p, dt / (p * steps) * 1e3, "equal" if len(set(out)) == 1 else out))
