"""Time of one zero-preserving list filter pass (lsf_convolve_axis_listed) at 256^3 on the sphere pair's band list, for 1, 2
and 3 planes: does the pass cost per vector-memory instruction (8 + 1 + 1 per plane and voxel) or per byte?
usage: list_pass_time.py [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c, l = sphere_pair(n, 3, "cuda")
grid = dev.make_grid((n, n, n))
band = dev.band_list(l, c, grid, _lib.BAND_ALL)
k7 = lsf.generate_1d_sobolev_kernel(7, 0.1)
for planes in (1, 2, 3):
    src = torch.randn((planes, n, n, n), device="cuda")
    dst = torch.zeros_like(src)
    for axis in (0, 1, 2):
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                dev.convolve_axis(src, dst, src, grid, axis, k7, None, band)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 100.0
            best = t if best is None else min(best, t)
        print("planes %d axis %d: %.1f us per pass (%d band voxels)" % (planes, axis, best, band.count))
