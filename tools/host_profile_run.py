import sys, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch, gc
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.synthetic import sphere_pair
n = 256
canonical, live0 = sphere_pair(n, 3, "cuda")
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING, maximum_warp_length_lower_threshold=0.0,
                               max_iterations=50, min_iterations=50, check_interval=50)
live = torch.empty_like(live0)
for _ in range(10):
    live.copy_(live0); opt.optimize(live, canonical)
gc.collect(); gc.freeze(); gc.disable()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    live.copy_(live0); opt.optimize(live, canonical)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
