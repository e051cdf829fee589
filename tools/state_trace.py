"""Shader-clock timeline of the INTERIOR list walk of the fused Slavcheva kernel (needs the -DLSF_STATE_TRACE build:
tools/build_variant.sh trace - -DLSF_STATE_TRACE).  Per traced wave and unit: cycles spent issuing the neighbourhood
loads, finishing the previous unit, waiting for the loads, in the arithmetic.  usage: state_trace.py [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("LSF_HIP_LIBRARY", os.path.join(ROOT, "levelsetfusion-python_amd/lib/variants/trace.so"))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BLOCKS, WAVES, UNITS, STAMPS = 64, 16, 16, 8
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
grid = dev.make_grid((n, n, n))
c, l = sphere_pair(n, 3, "cuda")
bands = dev.band_lists(l, c, grid)
rec = dev.new_records(1, "cuda")
st = dev.state_pack(l, None, grid, copies=2)
trace = torch.zeros(BLOCKS * WAVES * UNITS * STAMPS, dtype=torch.int64, device="cuda")
fn = _lib.lib.lsf_debug_set_state_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
waves = torch.zeros(4096 * 16 * 8, dtype=torch.int64, device="cuda")
for i in range(6):
    for b in bands:
        dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, b)
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(trace.data_ptr()), ctypes.c_void_p(waves.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(3):  # the third launch is the one looked at (stamps are overwritten)
    dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, bands[0])
e1.record()
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(0), ctypes.c_void_p(0)) == 0
wv = waves.cpu().numpy().reshape(4096, 16, 8)
used = wv[..., 0] != 0
nb = int(used[:, 0].sum())
wv = wv[:nb]
r0 = wv[..., 0].min()
ent, ls, le, ex = [(wv[..., k] - r0) * 10.0 for k in range(4)]  # ns at 100 MHz
clk = (wv[..., 5] - wv[..., 4]) / np.maximum((wv[..., 2] - wv[..., 1]) * 10e-9, 1e-9) / 1e9
print("%d blocks; kernel entry of the first wave = 0 ns" % nb)
print("entry      ns: min %.0f mean %.0f max %.0f" % (ent.min(), ent.mean(), ent.max()))
print("loop start ns: min %.0f mean %.0f max %.0f" % (ls.min(), ls.mean(), ls.max()))
print("loop end   ns: min %.0f mean %.0f max %.0f" % (le.min(), le.mean(), le.max()))
print("exit       ns: min %.0f mean %.0f max %.0f" % (ex.min(), ex.mean(), ex.max()))
print("loop duration ns: min %.0f mean %.0f max %.0f; in-kernel clock GHz: median %.2f (min %.2f max %.2f)" % (
    (le - ls).min(), (le - ls).mean(), (le - ls).max(), float(np.median(clk)), clk.min(), clk.max()))
xcc = wv[:, 0, 7] & 15
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: %3d blocks, entry %.0f..%.0f, loop end %.0f..%.0f, exit max %.0f" % (
            x, int(m.sum()), ent[m].min(), ent[m].max(), le[m].min(), le[m].max(), ex[m].max()))
hist, edges = np.histogram(ent[:, 0], bins=10)
print("entry histogram (ns):", list(zip(edges[:-1].astype(int), hist)))
t = trace.cpu().numpy().reshape(BLOCKS, WAVES, UNITS, STAMPS)
print("3 launches %.1f us, list %d entries, %d blocks" % (e0.elapsed_time(e1) * 1e3, bands[0].count, -1))
valid = t[..., 0] != 0
t0 = t[..., 0][valid].min()
names = ["issue loads", "finish prev", "(sched)", "wait loads", "arithmetic", "rest of unit"]
print("units traced per wave:", valid.sum(axis=2).ravel()[:16])
for b in (0, 9):
    for w in (0, 1, 4, 15):
        hw = int(t[b, w, 0, 6])
        print("block %d wave %d  HW_ID %#x (wave %d simd %d cu %d sh %d se %d) xcc %d" % (
            b, w, hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, int(t[b, w, 0, 7]) & 15))
        for u in range(UNITS):
            if not valid[b, w, u]:
                continue
            s = t[b, w, u]
            d = [int(s[1] - s[0]), int(s[2] - s[1]), int(s[3] - s[2]), int(s[4] - s[3]), int(s[5] - s[4])]
            nxt = int(t[b, w, u + 1, 0] - s[5]) if u + 1 < UNITS and valid[b, w, u + 1] else -1
            print("   unit %2d start %8d | issue %5d finish %5d wait %6d arith %6d tail %5d | to next %6d" % (
                u, int(s[0] - t0), *d, nxt))
# aggregate over all traced waves
d = np.stack([t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3],
              t[..., 5] - t[..., 4]], axis=-1)[valid]
print("means over %d wave-units: issue %.0f finish %.0f wait %.0f arith %.0f tail %.0f cycles" % ((len(d),) + tuple(d.mean(axis=0))))
span = (t[..., 5].max(axis=2) - np.where(valid, t[..., 0], np.iinfo(np.int64).max).min(axis=2))
print("per-wave span (first unit start -> last unit end): mean %.0f min %d max %d cycles; kernel span over traced waves %d"
      % (span.mean(), span.min(), span.max(), int(t[..., 5].max() - t0)))
out = os.path.join(ROOT, "gpurun_out", "state_trace_%d.npz" % n)
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, units=t, waves=wv)
print("saved", out)
