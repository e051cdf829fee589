#!/usr/bin/env python3
"""Host side of BASELINE config 2 (2-D 512^2 HierarchicalOptimizer2d, 3 levels, Tikhonov, 100 fixed iterations per level):
wall time per optimize(), a cProfile of one call, and -- with TRACE=1 under rocprofv3 --kernel-trace -- nothing else (the
trace is the profiler's).  Usage: host_profile_hier2d.py [size] [iterations] [blocked 0/1]"""
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
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
blocked = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
canonical, live0 = sphere_pair(n, 2, torch.device("cuda", 0))
opt = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
                                  maximum_iteration_count=iters, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05,
                                  check_interval=iters, engine_options=dict(blocked_levels=blocked))


def step():
    opt.optimize(canonical, live0)
    torch.cuda.synchronize()


for _ in range(5):
    t0 = time.perf_counter()
    step()
    print("blocked %s: step %.3f ms = %.2f us per iteration" % (blocked, (time.perf_counter() - t0) * 1e3,
                                                              (time.perf_counter() - t0) * 1e6 / (3 * iters)))
if os.environ.get("TRACE") != "1":
    pr = cProfile.Profile()
    pr.enable()
    step()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
