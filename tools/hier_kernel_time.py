import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair
n = 256
c, l = sphere_pair(n, 3, "cuda")
packed = dev.pack_live_gradient(l)
grid = dev.make_grid((n, n, n))
rec = dev.new_records(1, "cuda")
warp = torch.zeros((3, n, n, n), device="cuda")
g = [torch.zeros_like(warp), torch.zeros_like(warp)]
params = _lib.HierParams(1.0, 0.05, 0.1, 1, 1, 0)
for i in range(3):
    dev.hier_iteration(packed, c, warp, g[i % 2], g[(i + 1) % 2], grid, params, None, rec, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for i in range(20):
    dev.hier_iteration(packed, c, warp, g[i % 2], g[(i + 1) % 2], grid, params, None, rec, 0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("hier_iteration<TIK,UPDATE> 256^3: %.4f ms  %.2f TB/s (68 B/voxel)" % (ms, 68 * n ** 3 / ms / 1e9))
