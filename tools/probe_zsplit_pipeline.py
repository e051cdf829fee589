"""Probe: can the fixed part of a list launch (ramp-up + the drain of each wave's last unit, ~17 % of 30 us at 256^3) be
overlapped WITHOUT device-side synchronisation, by stream dependencies alone?  The band list is cut by z into a lower part
A, an upper part B and a thin middle zone M (2 * R slices); iteration i of A reads version i of A and of M's slices only,
B likewise, M reads a few slices of both.  Three streams:
    A_i waits for A_(i-1) [stream order] and M_(i-1) [event];  B_i likewise;  M_i waits for A_(i-1), B_(i-1), M_(i-1)
so A's workgroups of iteration i + 1 can start on CUs that B's of iteration i have not reached yet, and the drain of one
launch is covered by the ramp of the next.  The whole fixed-count run is captured in ONE HIP graph (no host cost) and
compared with the plain chain of launches captured the same way; the final states must be bit-identical.
usage: probe_zsplit_pipeline.py [size] [iterations] [R]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib

import torch

import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
grid = dev.make_grid((n, n, n))
c, l = sphere_pair(n, 3, "cuda")
bands = [b for b in dev.band_lists(l, c, grid) if b.count]
assert len(bands) == 1 and bands[0].subset == _lib.BAND_INTERIOR
whole = bands[0]
idx = whole.indices[:whole.count]
# cut where half of the band voxels lie below
mid_voxel = int(idx[whole.count // 2].item())
m = mid_voxel // (n * n)
keys = torch.tensor([(m - R) * n * n, (m + R) * n * n], dtype=torch.int32, device="cuda")
lo, hi = torch.searchsorted(idx, keys).tolist()
parts = {"A": dev.BandList(idx[:lo].contiguous(), lo, whole.subset),
         "M": dev.BandList(idx[lo:hi].contiguous(), hi - lo, whole.subset),
         "B": dev.BandList(idx[hi:].contiguous(), whole.count - hi, whole.subset)}
print("%d^3: band %d voxels; cut at slice %d +- %d: A %d, M %d, B %d" % (n, whole.count, m, R, lo, hi - lo,
                                                                           whole.count - hi), flush=True)
rec = dev.new_records(iters, "cuda")


def launch(st, i, band):
    dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, i, band)


def plain(st):
    for i in range(iters):
        launch(st, i, whole)


streams = {k: torch.cuda.Stream() for k in "ABM"}


def pipelined(st):
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event()
    fork.record(main)
    done = {k: None for k in "ABM"}  # the event behind part k's launch of the previous iteration
    for k in "ABM":
        streams[k].wait_event(fork)
    for i in range(iters):
        now = {}
        for k, needs in (("A", "M"), ("B", "M"), ("M", "AB")):
            s = streams[k]
            for other in needs:
                if done[other] is not None:
                    s.wait_event(done[other])
            with torch.cuda.stream(s):
                launch(st, i, parts[k])
                e = torch.cuda.Event()
                e.record(s)
            now[k] = e
        done = now
    for k in "ABM":
        main.wait_event(done[k])


import ctypes

helper = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "bin", "libzsplit_graph.so"))
helper.zsplit_build.restype = ctypes.c_void_p
helper.zsplit_build.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_int32, ctypes.c_int32]
helper.zsplit_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
fn = ctypes.cast(_lib.lib.lsf_slavcheva_state_iteration, ctypes.c_void_p)


def run(mode):
    """mode: "plain" / "split" as HIP graphs built by the helper (stream capture in C++), "eager-split" from Python"""
    st = dev.state_pack(l, None, grid, copies=2)
    st2 = dev.state_pack(l, None, grid, copies=2)
    rec.zero_()
    plain(st)  # warm-up outside capture
    torch.cuda.synchronize()
    plan = None
    if mode != "eager-split":
        order = ["A", "B", "M"] if mode == "split" else None
        lists = [parts[k] for k in order] if order else [whole, whole, whole]
        arr = (ctypes.c_void_p * 3)(*[b.indices.data_ptr() for b in lists])
        cnt = (ctypes.c_int64 * 3)(*[b.count for b in lists])
        plan = helper.zsplit_build(fn, st[0].data_ptr(), st[1].data_ptr(), c.data_ptr(), ctypes.addressof(grid),
                                   ctypes.addressof(eng.params), rec.data_ptr(), _lib.RECORD_BYTES, iters,
                                   ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), whole.subset,
                                   1 if mode == "split" else 0)
        assert plan, "graph capture failed"
    best = 1e9
    for rep in range(6):
        for a, b in zip(st, st2):
            a.copy_(b)
        rec.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        if plan:
            assert helper.zsplit_launch(plan, dev.stream_ptr()) == 0
        else:
            pipelined(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    final = st[iters % 2]
    h = hashlib.sha1(final.cpu().numpy().tobytes()).hexdigest()[:16]
    host = dev.decode_records(dev.records_to_host(rec))
    return best, h, float(host["max_value"][-1]), float(host["data_energy"][-1])


import time


def run_eager_cpp():
    """the same dependencies enqueued by the helper WITHOUT a graph (host: ~10 HIP calls per iteration from C++); wall time
    of enqueue + execution, the helper synchronises at the end"""
    st = dev.state_pack(l, None, grid, copies=2)
    st2 = dev.state_pack(l, None, grid, copies=2)
    lists = [parts[k] for k in ("A", "B", "M")]
    arr = (ctypes.c_void_p * 3)(*[b.indices.data_ptr() for b in lists])
    cnt = (ctypes.c_int64 * 3)(*[b.count for b in lists])
    best = 1e9
    for rep in range(6):
        for a, b in zip(st, st2):
            a.copy_(b)
        rec.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = helper.zsplit_build(fn, st[0].data_ptr(), st[1].data_ptr(), c.data_ptr(), ctypes.addressof(grid),
                                   ctypes.addressof(eng.params), rec.data_ptr(), _lib.RECORD_BYTES, iters,
                                   ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), whole.subset, 2)
        dt = time.perf_counter() - t0
        assert plan
        best = min(best, dt / iters * 1e6)
    final = st[iters % 2]
    h = hashlib.sha1(final.cpu().numpy().tobytes()).hexdigest()[:16]
    host = dev.decode_records(dev.records_to_host(rec))
    return best, h, float(host["max_value"][-1]), float(host["data_energy"][-1])


modes = [("one launch per iteration (graph)", "plain"), ("A | M | B, enqueued from Python", "eager-split")]
if os.environ.get("SPLIT_GRAPH") == "1":  # multi-stream capture: the HIP runtime of this image crashes in it
    modes.append(("A | M | B on three streams (graph)", "split"))
for name, mode in modes:
    us, h, mx, en = run(mode)
    print("%-36s %.2f us per iteration   state %s  last max %.10f  data energy %.9f" % (name, us, h, mx, en), flush=True)
us, h, mx, en = run_eager_cpp()
print("%-36s %.2f us per iteration   state %s  last max %.10f  data energy %.9f (wall clock, enqueue included)"
      % ("A | M | B, enqueued from C++", us, h, mx, en), flush=True)
