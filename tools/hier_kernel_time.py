#!/usr/bin/env python3
"""HIP-event time of lsf_hier_iteration alone (3-D, Tikhonov, with / without the in-kernel update) on a sphere pair.
Usage: hier_kernel_time.py [size] [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from levelsetfusion_python_amd import _lib, device as dev  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
canonical, live = sphere_pair(n, 3, "cuda")
packed = dev.pack_live_gradient(live)
grid = dev.make_grid((n, n, n))
warp = torch.zeros((3, n, n, n), device="cuda")
F = [torch.zeros_like(warp) for _ in range(2)]
rec = dev.new_records(2, warp.device)
for update in (1, 0):
    params = _lib.HierParams(1.0, 0.2, 0.1, 1, update, 0)
    warp.zero_()
    for k in range(3):
        dev.hier_iteration(packed, canonical, warp, F[k % 2], F[(k + 1) % 2], grid, params, None, rec, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(reps):
        dev.hier_iteration(packed, canonical, warp, F[k % 2], F[(k + 1) % 2], grid, params, None, rec, 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = (68 if update else 56) * n ** 3
    print("%d^3 hier_iteration (Tikhonov, update=%d): %.4f ms = %.0f GB/s of %d B per voxel"
          % (n, update, ms, nbytes / ms / 1e6, 68 if update else 56))
