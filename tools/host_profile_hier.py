#!/usr/bin/env python3
"""cProfile of one HierarchicalOptimizer3d.optimize() call (host side).  Usage: host_profile_hier.py [size] [full 0/1]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
full = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
canonical, live0 = sphere_pair(n, 3, torch.device("cuda", 0))
k7 = lsf.generate_1d_sobolev_kernel(7, 0.1)
opt = lsf.HierarchicalOptimizer3d(tikhonov_term_enabled=True, gradient_kernel_enabled=full, maximum_chunk_size=8, rate=0.1,
                                  maximum_iteration_count=50, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05,
                                  kernel=k7 if full else None, check_interval=50)


def step():
    opt.optimize(canonical, live0)
    torch.cuda.synchronize()


for _ in range(3):
    t0 = time.perf_counter()
    step()
    print("step %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
