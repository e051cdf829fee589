#!/usr/bin/env python3
"""cProfile of one SlavchevaOptimizer3d.optimize() call (host side) at a given volume size: shows where the host
blocks (allocations, synchronising copies) next to the enqueue cost.  Usage: host_profile.py [size] [iterations]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
device = torch.device("cuda", 0)
canonical, live0 = sphere_pair(n, 3, device)
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                               maximum_warp_length_lower_threshold=0.0, max_iterations=iters, min_iterations=iters,
                               check_interval=iters)
live = torch.empty_like(live0)


def step():
    live.copy_(live0)
    opt.optimize(live, canonical)
    torch.cuda.synchronize()


for _ in range(2):
    t0 = time.perf_counter()
    step()
    print("step %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
