"""Known-traffic launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in THIS code's access patterns:
  4 bytes per lane, unit stride: hier_update_kernel reads g (12 B) + warp (12 B) and writes warp (12 B) per voxel at 256^3
      -> 402 653 184 B read, 201 326 592 B written per launch;
  16 bytes per lane (the float4 state): state_unpack_kernel reads 16 B per voxel -> 268 435 456 B read (and writes
      4 + 12 B as dword streams); state_pack_kernel reads 4 B and writes 2 x 16 B per voxel -> 536 870 912 B written.
(buffers >> 256 MiB Infinity Cache)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd  # noqa: F401
from levelsetfusion_python_amd import device as dev

n = 256
g = torch.randn((3, n, n, n), device="cuda") * 1e-3
w = torch.zeros((3, n, n, n), device="cuda")
rec = dev.new_records(1, "cuda")
grid = dev.make_grid((n, n, n))
for _ in range(10):
    dev.hier_update(g, w, grid, 0.1, None, rec, 0)
live = torch.rand((n, n, n), device="cuda")
out_live = torch.empty_like(live)
for _ in range(10):
    a, b = dev.state_pack(live, None, grid, copies=2)
    dev.state_unpack(a, grid, out_live, w, None)
torch.cuda.synchronize()
print("calibration launches done")
