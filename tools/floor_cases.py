"""Launches ONE variant of the fused KillingFusion iteration a few times on the sphere pair -- target of tools/floor_table.sh's
rocprofv3 --pmc passes (round 6: what binds the kernel, per variant).  Environment: N (256), WALK = list | box, ENERGY = 1 | 0
(0: the three energy sums left out -- not a product configuration: the reference logs them every iteration,
slavcheva_optimizer2d.py:371-374), LAUNCHES (12)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd import _lib, device as dev  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(os.environ.get("N", "256"))
walk = os.environ.get("WALK", "list")
launches = int(os.environ.get("LAUNCHES", "12"))
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
if os.environ.get("ENERGY", "1") == "0":
    eng.params.energy_mode = _lib.ENERGY_NONE
canonical, live = sphere_pair(n, 3, "cuda")
grid = dev.make_grid((n, n, n))
prepared = dev.StatePrepare(live, canonical, grid, sparse_reach=2)
bands, _ = prepared.collect()
interior = [b for b in bands if b.subset == _lib.BAND_INTERIOR and b.count]
records = dev.new_records(launches, "cuda")
states = prepared.states
if walk == "box":
    boxes, n_boxes = dev.band_boxes(prepared)
    canonical_boxed = dev.band_boxes_canonical(canonical, grid, boxes, n_boxes)
start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(launches):
    if i == 2:
        start.record()
    if walk == "box":
        dev.slavcheva_state_iteration_boxes(states[i % 2], canonical_boxed, states[(i + 1) % 2], grid, eng.params, None,
                                            records, i, boxes, n_boxes)
    else:
        for b in interior:
            dev.slavcheva_state_iteration(states[i % 2], canonical, states[(i + 1) % 2], grid, eng.params, None, records, i, b)
stop.record()
torch.cuda.synchronize()
print("N %d walk %s energy %s: %d band voxels, %.2f us per launch (HIP events over %d back-to-back launches)"
      % (n, walk, os.environ.get("ENERGY", "1"), sum(b.count for b in interior), start.elapsed_time(stop) * 1e3 / (launches - 2),
         launches - 2), flush=True)
