import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair
n = 256
ct, lt = sphere_pair(n, 3, "cuda")
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True, smoothing_term_method=lsf.SmoothingTermMethod.KILLING)._engine
warp_prev = (0.3 * torch.randn((3, n, n, n), device="cuda", generator=torch.Generator("cuda").manual_seed(3)))
rec = dev.new_records(4, "cuda")
outs = []
for rep in range(2):
    full_w, full_l = torch.empty_like(warp_prev), torch.empty_like(lt)
    dev.slavcheva_iteration(_lib.STAGE_FUSED, lt, ct, warp_prev, full_w, full_l, None, dev.make_grid(lt.shape), eng.params, None, rec, 0)
    outs.append((full_w, full_l))
print("full repeat equal:", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[0][1], outs[1][1]))
full_w, full_l = outs[0]
print("max |warp|", float(full_w.abs().max()))
h, half = 2, n // 2
for k, (a, b, zb, ze) in enumerate(((0, half + h, 0, half), (half - h, n, h, h + half))):
    ls, cs, ws = lt[a:b].contiguous(), ct[a:b].contiguous(), warp_prev[:, a:b].contiguous()
    ow, ol = torch.zeros_like(ws), torch.zeros_like(ls)
    dev.slavcheva_iteration(_lib.STAGE_FUSED, ls, cs, ws, ow, ol, None, dev.make_grid(ls.shape, zb, ze, a), eng.params, None, rec, 1 + k)
    dl = (ol[zb:ze] != full_l[a + zb:a + ze])
    dw = (ow[:, zb:ze] != full_w[:, a + zb:a + ze])
    print("slab", k, "live mismatches", int(dl.sum()), "warp mismatches", int(dw.sum()))
    if dl.any():
        idx = dl.nonzero()
        print(" z range of mismatches (local owned idx):", int(idx[:,0].min()), int(idx[:,0].max()), "count per z:", torch.bincount(idx[:,0])[:8].tolist())
        z,y,x = idx[0].tolist()
        print(" first", (z,y,x), float(ol[zb+z,y,x]), float(full_l[a+zb+z,y,x]), "warp z", float(full_w[2,a+zb+z,y,x]))
