"""One iteration of the fused Slavcheva kernel on a band that touches the volume's faces (the synthetic depth pair): the
INTERIOR list on the CU-sized walk + the BOUNDARY list on the general walk (two launches: what the engine does) against ONE
launch of the general walk over the whole band.  usage: list_split_modes.py [sizes]  (default 256,512)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import depth_pair, sphere_pair

iters = 50
for kind in ("depth", "sphere"):
    for n in [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "256,512").split(",")]:
        eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
        grid = dev.make_grid((n, n, n))
        c, l = depth_pair(n, "cuda") if kind == "depth" else sphere_pair(n, 3, "cuda")
        rec = dev.new_records(iters, "cuda")
        modes = {"interior + boundary": dev.band_lists(l, c, grid, split=True),
                 "one general launch": dev.band_lists(l, c, grid, split=False)}
        for name, bands in modes.items():
            bands = [b for b in bands if b.count]
            best = 1e9
            for rep in range(5):
                st = dev.state_pack(l, None, grid, copies=2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(iters):
                    for b in bands:
                        dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, b)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters * 1e3)
            print("%-6s %4d^3  %-22s lists %s: %.2f us per iteration" % (kind, n, name, [b.count for b in bands], best),
                  flush=True)
