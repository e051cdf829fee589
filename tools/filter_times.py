#!/usr/bin/env python3
"""3-D Sobolev filter of a planar 3-vector field: lsf_convolve_xyz (one launch) against the three lsf_convolve_axis
passes, HIP-event times per call and the equality of the results.  Usage: filter_times.py [size] [taps] [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd import device as dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_taps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
taps = lsf.generate_1d_sobolev_kernel(n_taps, 0.1)
grid = dev.make_grid((n, n, n))
src = torch.randn((3, n, n, n), device="cuda")
a, b, fused = torch.empty_like(src), torch.empty_like(src), torch.empty_like(src)


def three():
    dev.convolve_axis(src, a, None, grid, 0, taps)
    dev.convolve_axis(a, b, None, grid, 1, taps)
    dev.convolve_axis(b, a, None, grid, 2, taps)


def one():
    dev.convolve_xyz(src, fused, grid, taps)


def timed(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t3, t1 = timed(three), timed(one)
gb = 3 * n ** 3 * 4 / 1e9
print("%d^3, %d taps: three passes %.3f ms (%.0f GB/s of 6 field transfers), fused %.3f ms (%.0f GB/s of 2), equal: %s"
      % (n, n_taps, t3, 6 * gb / t3 * 1e3, t1, 2 * gb / t1 * 1e3, bool(torch.equal(a, fused))))
