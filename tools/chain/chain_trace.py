"""100 MHz timeline of the chain kernel's work items (built here with -DLSF_CHAIN_TRACE).  Per workgroup and item: when it began to wait for its window,
when the poll matched, when the acquire had completed, when the walk began, when every wave had drained, when the item
was published.  usage: chain_trace.py [size] [iterations] [stages]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
import chain as chain_tool

# the trace build of the tool's own library (-DLSF_CHAIN_TRACE)
os.environ["LSF_CHAIN_LIBRARY"] = chain_tool.build(("-DLSF_CHAIN_TRACE",), os.path.join(os.path.dirname(chain_tool.LIB_PATH), "liblsf_chain_trace.so"))
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ITEMS, WORDS = 64, 32
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
grid = dev.make_grid((n, n, n))
c, l = sphere_pair(n, 3, "cuda")
bands = dev.band_lists(l, c, grid)
assert len(bands) == 1
rec = dev.new_records(iters, "cuda")
st = dev.state_pack(l, None, grid, copies=2)
chain = chain_tool.StateChain(st, c, grid, eng.params, rec, bands[0], stages)
print("workgroups %d stages %d chunks %d units %d" % (chain.workgroups, chain.stages_used, chain.chunks, chain.units))
fn = chain_tool.chain_lib().lsf_debug_set_chain_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
trace = torch.zeros(chain.workgroups * ITEMS * WORDS, dtype=torch.int64, device="cuda")
chain.launch(0, iters)
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(trace.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
chain.launch(0, iters)
e1.record()
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(0)) == 0
print("launch %.1f us = %.2f us per iteration (with stamps)" % (e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / iters))
t = trace.cpu().numpy().reshape(chain.workgroups, ITEMS, WORDS)
items_per_wg = int((t[0, :, 0] != 0).sum())
t = t[:, :items_per_wg]
t0 = t[:, :, 0].min()
us = lambda x: (x - t0) / 100.0
begin, matched, acquired, walk, drained, published = (us(t[:, :, k]) for k in (0, 1, 2, 3, 4, 5))
it = t[:, :, 7] >> 32
chunk = t[:, :, 7] & 0xffffffff
wave_end = us(t[:, :, 16:32])
print("items per workgroup: %d" % items_per_wg)
first = begin[:, 0]
print("first item begins: min %.2f max %.2f us" % (first.min(), first.max()))
waited = it > 0
wait = np.where(waited, matched - begin, 0.0)
acq = np.where(waited, acquired - matched, 0.0)
body = drained - walk
tail = drained - wave_end.min(axis=2)
print("per item, us (mean / p50 / p90 / max):")
for name, v, m in (("wait for window", wait, waited), ("acquire", acq, waited), ("barrier after", walk - np.where(waited, acquired, begin), None),
                   ("walk (start -> all drained)", body, None), ("first wave done -> all drained", tail, None),
                   ("reduce + publish", published - drained, None), ("item total", published - begin, None)):
    x = v[m] if m is not None else v.ravel()
    print("  %-32s %6.2f %6.2f %6.2f %6.2f" % (name, x.mean(), np.median(x), np.percentile(x, 90), x.max()))
per_it = []
for i in range(int(it.max()) + 1):
    m = it == i
    per_it.append((published[m].max(), begin[m].min(), body[m].mean(), wait[m].mean()))
print("iteration: last publish, first begin, mean walk, mean wait (us)")
for i, row in enumerate(per_it[:24]):
    print("  %2d  %8.2f %8.2f %6.2f %6.2f" % ((i,) + row))
if len(per_it) > 4:
    ends = np.array([r[0] for r in per_it])
    print("steady state: %.2f us per iteration (last publish, iterations 2..)" % ((ends[-1] - ends[1]) / (len(ends) - 2)))
# who is slow: mean walk time per workgroup, by XCD
xcc = t[:, 0, 8] & 15
wg_walk = body.mean(axis=1)
print("walk time per workgroup: min %.2f mean %.2f max %.2f us" % (wg_walk.min(), wg_walk.mean(), wg_walk.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCD %d: %3d workgroups, walk mean %.2f max %.2f, wait mean %.2f" % (x, m.sum(), wg_walk[m].mean(), wg_walk[m].max(),
                                                                              wait[m].mean()))
polls = t[:, :, 9]
print("unmatched polls per waited item: mean %.1f max %d" % (polls[waited].mean(), polls.max()))
