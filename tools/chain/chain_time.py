"""Time of the chain kernel per iteration (HIP events around ONE launch of `iters` iterations, best of 5) next to the
per-iteration kernel's, and a bit-for-bit comparison of the two final states.  The chain's geometry comes from the
environment (LSF_CHAIN_THREADS, LSF_CHAIN_BLOCKS), one configuration per process.
usage: chain_time.py [size] [iterations] [stages]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import device as dev
import chain as chain_tool
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
grid = dev.make_grid((n, n, n))
c, l = sphere_pair(n, 3, "cuda")
bands = dev.band_lists(l, c, grid)
assert len(bands) == 1


def timed(run):
    best = None
    for _ in range(5):
        st = dev.state_pack(l, None, grid, copies=2)
        rec = dev.new_records(iters, "cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prepared = run(st, rec, None)
        torch.cuda.synchronize()
        e0.record()
        run(st, rec, prepared)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / iters
        best = t if best is None else min(best, t)
    return best, st, rec


def run_chain(st, rec, chain):
    if chain is None:
        return chain_tool.StateChain(st, c, grid, eng.params, rec, bands[0], stages)
    assert chain.launch(0, iters)
    return chain


def run_launches(st, rec, go):
    if go is None:
        return True
    for i in range(iters):
        dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, i, bands[0])
    return True


t_chain, st_a, rec_a = timed(run_chain)
t_each, st_b, rec_b = timed(run_launches)
same = all(torch.equal(a, b) for a, b in zip(st_a, st_b))
da, db = dev.decode_records(dev.records_to_host(rec_a)), dev.decode_records(dev.records_to_host(rec_b))
same_rec = (da["max_value"] == db["max_value"]).all() and (da["argmax"] == db["argmax"]).all()
count = bands[0].count
probe = chain_tool.StateChain(st_a, c, grid, eng.params, rec_a, bands[0], stages)
print("%d^3 threads=%s stages=%d(%d) wg=%d chunks=%d: chain %.2f us/iteration (%.3f of the HBM roofline at 52 B), per-iteration "
      "launches %.2f us (%.3f); states equal %s, records equal %s" % (
          n, os.environ.get("LSF_CHAIN_THREADS", "default"), stages, probe.stages_used, probe.workgroups, probe.chunks, t_chain,
          52.0 * count / (t_chain * 1e-6) / 8e12, t_each, 52.0 * count / (t_each * 1e-6) / 8e12, same, bool(same_rec)))
