"""HIP-event time of lsf_convolve_axis_listed4 (one zero-preserving filter pass on the float4 gradient layout) per axis at
256^3 on the sphere pair's band list.  (The persistent, software-pipelined variant DESIGN.md section 7 compares it with
was selected by LSF_SOBOLEV_PIPE = workgroups per XCD in a measurement build; it is not in the tree.)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair
n = 256
c, l = sphere_pair(n, 3, "cuda")
grid = dev.make_grid((n, n, n))
band = dev.band_list(l, c, grid, _lib.BAND_ALL)
k7 = np.ascontiguousarray(np.asarray(lsf.generate_1d_sobolev_kernel(7, 0.1), dtype=np.float64))
src = torch.randn((n, n, n, 4), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
outs = {}
for axis in (0, 1, 2):
    dst = torch.zeros_like(src)
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            _lib.check(_lib.lib.lsf_convolve_axis_listed4(src.data_ptr(), dst.data_ptr(), src.data_ptr(), ctypes.byref(grid), axis,
                       k7.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 7, None, band.pointer, band.count, dev.stream_ptr()), "x")
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 100.0
        best = t if best is None else min(best, t)
    print("axis %d: %.1f us per pass" % (axis, best))
