"""One box, alternating: BASELINE config 4 with the whole call enqueued by the library (engine.library_run = True:
lsf_state_run_begin / _finish) and launch by launch from Python (False), milliseconds per optimize() over `steps` calls,
three rounds each; final live fields must be equal.  usage: run_path_ab.py [size] [steps]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
canonical, live0 = sphere_pair(n, 3, "cuda")
opts = {}
for run in (True, False):
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                   maximum_warp_length_lower_threshold=0.0, max_iterations=50, min_iterations=50,
                                   check_interval=50)
    opt.engine.library_run = run
    opts[run] = opt
live = torch.empty_like(live0)
finals = {}
gc.collect()
gc.freeze()
gc.disable()
for rnd in range(3):
    for run in (True, False):
        opt = opts[run]
        for _ in range(5):
            live.copy_(live0)
            opt.optimize(live, canonical)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            live.copy_(live0)
            opt.optimize(live, canonical)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        finals[run] = live.clone()
        print("%d^3 round %d  %-34s %.4f ms per optimize()" % (n, rnd, "enqueued by the library" if run else
                                                             "launch by launch from Python", dt))
print("final live fields equal:", torch.equal(finals[True], finals[False]))
