"""Fused kernel time at 256^3 on three inputs: nothing in the narrow band, the sphere pair (~10 % in band), everything
in the band -- separates the floor (streaming 36 B/voxel) from the per-band-voxel cost and shows load balance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
if os.environ.get("ENERGY", "1") == "0":
    eng.params.energy_mode = _lib.ENERGY_NONE
grid = dev.make_grid((n, n, n))
rec = dev.new_records(128, "cuda")
c_s, l_s = sphere_pair(n, 3, "cuda")
ones = torch.ones_like(c_s)
z, y, x = torch.meshgrid(*[torch.arange(n, device="cuda", dtype=torch.float32)] * 3, indexing="ij")
ramp_c = (0.8 * torch.sin(x * 0.05) * torch.cos(y * 0.04) * torch.cos(z * 0.03)).contiguous()
ramp_l = (0.8 * torch.sin(x * 0.05 + 0.1) * torch.cos(y * 0.04 - 0.05) * torch.cos(z * 0.03 + 0.08)).contiguous()
for name, c, l in (("none in band", ones, ones.clone()), ("sphere pair", c_s, l_s), ("all in band", ramp_c, ramp_l)):
    band = float((~((l.abs() == 1) & (c.abs() == 1))).float().mean())
    for listed in (False, True):
        st = dev.state_pack(l, None, grid, copies=2)
        bands = dev.band_lists(l, c, grid, split=os.environ.get("SPLIT", "1") == "1") if listed else [None]
        for i in range(4):
            for b in bands:
                dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 20
        for i in range(reps):
            for b in bands:
                dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-14s band fraction %.3f  band list %-5s %.4f ms  -> %.1f G voxel-updates/s"
              % (name, band, listed, ms, n ** 3 / ms / 1e6))
