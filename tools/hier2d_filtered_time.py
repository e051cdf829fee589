#!/usr/bin/env python3
"""wall time per iteration of HierarchicalOptimizer2d WITH the gradient kernel (the reference's default constructor:
hierarchical_optimizer2d.py:63-73) on a 2-D sphere pair.  usage: hier2d_filtered_time.py [size] [iterations] [calls]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
canonical, live = sphere_pair(n, 2, "cuda")
CASES = (("(warm-up)", dict(gradient_kernel_enabled=True, maximum_warp_update_threshold=0.0)),
         ("Tikhonov + 7-tap kernel, threshold 0.001", dict(gradient_kernel_enabled=True, maximum_warp_update_threshold=0.001)),
         ("Tikhonov + 7-tap kernel, fixed count", dict(gradient_kernel_enabled=True, maximum_warp_update_threshold=0.0)),
         ("Tikhonov only, threshold 0.001", dict(gradient_kernel_enabled=False, maximum_warp_update_threshold=0.001)))
for blocked in (True, False):
  for label, kw in CASES:
    opt = lsf.HierarchicalOptimizer2d(engine_options=dict(blocked_levels=blocked), tikhonov_term_enabled=True, maximum_chunk_size=4, rate=0.1, maximum_iteration_count=iters,
                                      tikhonov_strength=0.05, kernel=lsf.generate_1d_sobolev_kernel(7, 0.1), **kw)
    for k in range(calls + 4):
        if k == 4:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.optimize(canonical, live)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    counts = opt.get_per_level_iteration_counts()
    if not label.startswith("("):
        print("%d^2 %-42s %-22s %7.3f ms per call, levels %s: %.2f us per iteration"
              % (n, label, "blocked levels" if blocked else "one launch per pass", elapsed / calls * 1e3, counts,
                 elapsed / calls / sum(counts) * 1e6))
