"""Shader-clock totals per wave of the SobolevFusion box kernel (needs the -DLSF_SOB_TRACE build: tools/build_variant.sh
sobtrace - -DLSF_SOB_TRACE): where a wave's rounds go -- issuing the staging loads, waiting for them (and for the previous
round's stores), the y pass, the z pass + update + re-warp, the stores.  usage: sobolev_trace.py [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("LSF_HIP_LIBRARY", os.path.join(ROOT, "levelsetfusion-python_amd/lib/variants/sobtrace.so"))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
canonical, live0 = sphere_pair(n, 3, "cuda")
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                               sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1),
                               maximum_warp_length_lower_threshold=0.0, max_iterations=10, min_iterations=10, check_interval=10)
fn = _lib.lib.lsf_debug_set_sobolev_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
rows = torch.zeros(256 * 16 * 8, dtype=torch.int64, device="cuda")
live = live0.clone()
opt.optimize(live, canonical)
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(rows.data_ptr())) == 0
live.copy_(live0)
opt.optimize(live, canonical)  # the last launch's totals stay
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(0)) == 0
r = rows.cpu().numpy().reshape(256, 16, 8).astype(np.float64)
used = r[..., 0] > 0
print("%d waves walked boxes; rounds per wave: min %d mean %.1f max %d" % (used.sum(), r[..., 0][used].min(), r[..., 0][used].mean(),
                                                                      r[..., 0][used].max()))
names = ["stage issue", "wait for the staged data", "y pass", "z pass + update + re-warp", "stores + loop"]
tot = r[..., 6][used]
print("cycles per wave, entry to exit: mean %.0f (max %.0f); per round %.0f" % (tot.mean(), tot.max(), (tot / r[..., 0][used]).mean()))
for k, name in enumerate(names):
    per = (r[..., 1 + k][used] / r[..., 0][used])
    print("  %-28s %7.0f cycles per round (%4.1f %% of a wave's time)" % (name, per.mean(), 100.0 * r[..., 1 + k][used].sum() / tot.sum()))
start = r[..., 7][used]
print("entry stamps (100 MHz): spread %.0f ns" % ((start.max() - start.min()) * 10.0))
