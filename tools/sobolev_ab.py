"""One box, alternating: the SobolevFusion call of bench.py --workload sobolev with everything behind the x pass box by box in
one launch (engine.sobolev_boxes = True: lsf_sobolev_state_update_boxes) and on lists (False: lsf_convolve_axis_listed4 +
lsf_sobolev_state_update), milliseconds per optimize() over `steps` calls, three rounds each; final live fields must be
equal.  usage: sobolev_ab.py [size] [steps] [boxes|lists: that side only]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sides = {"boxes": (True,), "lists": (False,)}.get(sys.argv[3] if len(sys.argv) > 3 else "", (True, False))
canonical, live0 = sphere_pair(n, 3, "cuda")
opts = {}
for boxes in sides:
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                                   sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1),
                                   maximum_warp_length_lower_threshold=0.0, max_iterations=50, min_iterations=50,
                                   check_interval=50)
    opt.engine.sobolev_boxes = boxes
    opts[boxes] = opt
live = torch.empty_like(live0)
finals = {}
gc.collect()
gc.freeze()
gc.disable()
for rnd in range(3):
    for boxes in sides:
        opt = opts[boxes]
        for _ in range(3):
            live.copy_(live0)
            opt.optimize(live, canonical)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            live.copy_(live0)
            opt.optimize(live, canonical)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        finals[boxes] = live.clone()
        band = opt.engine._sobolev_band.count
        print("%d^3 round %d  %-28s %.4f ms per optimize() = %.1f us per iteration, %.3f of the HBM roofline at 76 B per "
              "band voxel (%d); boxes used: %s" % (n, rnd, "boxes behind the x pass" if boxes else "lists", dt, dt * 20.0,
                                                  76.0 * band * 50 / (dt * 1e-3) / 8e12, band, opt.engine.last_call.sobolev_boxes))
if len(sides) == 2:
    print("final live fields equal:", torch.equal(finals[True], finals[False]))
