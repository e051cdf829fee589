#!/usr/bin/env python3
"""per-step wall time of SlavchevaOptimizer3d.optimize() (bench workload) over many steps: shows allocator warm-up,
clock drift and outliers.  Usage: step_times.py [size] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
iters = 50
device = torch.device("cuda", 0)
canonical, live0 = sphere_pair(n, 3, device)
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                               maximum_warp_length_lower_threshold=0.0, max_iterations=iters, min_iterations=iters,
                               check_interval=iters)
live = torch.empty_like(live0)
times = []
import gc
if os.environ.get("NOGC") == "1":
    gc.disable()
segs = []
for s in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    live.copy_(live0)
    opt.optimize(live, canonical)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    st = torch.cuda.memory_stats()
    segs.append((st["num_device_alloc"], st["num_device_free"], st["num_alloc_retries"]))
print(" ".join("%.2f" % t for t in times))
print(" ".join("%d/%d/%d" % t for t in segs[::4]))
print("reserved %.2f GB allocated %.2f GB" % (torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30))
