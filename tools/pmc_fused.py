"""launches the fused Slavcheva state kernel a few times on the 256^3 sphere pair: 6 x band lists (the bench's launches,
grid 256 x 1024), then 6 x the dense walk of the same pair (grid 2008 x 256) -- target for rocprofv3 --pmc passes
(tools/pmc_passes.sh; the summary separates the two by grid size).  CASES=all adds an all-in-band pair (lists + dense)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(os.environ.get("N", "256"))
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
if os.environ.get("ENERGY", "1") == "0":
    eng.params.energy_mode = _lib.ENERGY_NONE
grid = dev.make_grid((n, n, n))
rec = dev.new_records(1, "cuda")
c_s, l_s = sphere_pair(n, 3, "cuda")
z, y, x = torch.meshgrid(*[torch.arange(n, device="cuda", dtype=torch.float32)] * 3, indexing="ij")
ramp_c = (0.8 * torch.sin(x * 0.05) * torch.cos(y * 0.04) * torch.cos(z * 0.03)).contiguous()
ramp_l = (0.8 * torch.sin(x * 0.05 + 0.1) * torch.cos(y * 0.04 - 0.05) * torch.cos(z * 0.03 + 0.08)).contiguous()
cases = [(c_s, l_s, True), (c_s, l_s, False)]
if os.environ.get("CASES") == "all":
    cases += [(ramp_c, ramp_l, True), (ramp_c, ramp_l, False)]
for c, l, listed in cases:
    bands = dev.band_lists(l, c, grid, bytes_per_voxel=16) if listed else [None]
    st = dev.state_pack(l, None, grid)
    for i in range(6):
        for b in bands:
            dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, b)
torch.cuda.synchronize()
