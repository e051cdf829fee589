"""Throughput of independent pairs when P of them are in flight at once (the reference's multi-pair loop is embarrassingly
parallel: run_hierarchical_optimizer3d_multipair.py:403-432): P host threads, each with its own optimizer and HIP stream,
run `steps` optimize() calls on pairs of their own.  One iteration launch fills every CU with a 1024-thread workgroup, so a
second stream's launch moves into the CUs as the first one's workgroups retire: the drain of one launch and the gap to the
next are filled by the other pair.  usage: pairs_in_flight.py [size] [steps] [iterations]"""
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
canonical, live0 = sphere_pair(n, 3, "cuda")


def worker(stream, count, barrier, out, k):
    with torch.cuda.stream(stream):
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                       maximum_warp_length_lower_threshold=0.0, max_iterations=iters, min_iterations=iters,
                                       check_interval=iters)
        live = torch.empty_like(live0)
        for _ in range(3):
            live.copy_(live0)
            opt.optimize(live, canonical)
        stream.synchronize()
        barrier.wait()
        for _ in range(count):
            live.copy_(live0)
            opt.optimize(live, canonical)
        stream.synchronize()
        out[k] = float(live.double().sum().item())


for p in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(p)]
    barrier = threading.Barrier(p + 1)
    out = [None] * p
    threads = [threading.Thread(target=worker, args=(streams[k], steps, barrier, out, k)) for k in range(p)]
    for t in threads:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    print("%d pair(s) in flight: %.3f ms per pair, %.1f G voxel-updates/s (checksums %s)" % (
        p, dt / (p * steps) * 1e3, n ** 3 * iters * p * steps / dt / 1e9, "equal" if len(set(out)) == 1 else out))
