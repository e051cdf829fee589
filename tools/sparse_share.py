"""share of the 1024-voxel chunks the sparse state initialisation writes (dev.StatePrepare(sparse_reach=2)), sphere pair"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from levelsetfusion_python_amd import device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair, depth_pair
for n in (128, 192, 256, 384, 512):
    c, l = sphere_pair(n, 3, "cuda")
    band = float((~((l.abs() == 1) & (c.abs() == 1))).float().mean())
    out = []
    for reach in (1, 2):
        p = dev.StatePrepare(l, c, sparse_reach=reach)
        p.collect()
        out.append(p.needed_fraction())
    print("sphere pair %d^3: band %.1f %% of the voxels; chunks written with reach 1 / 2: %.1f %% / %.1f %%" % (n, 100 * band, 100 * out[0], 100 * out[1]))
for n in (256, 512):
    c, l = depth_pair(n, "cuda")
    band = float((~((l.abs() == 1) & (c.abs() == 1))).float().mean())
    p = dev.StatePrepare(l, c, sparse_reach=2)
    p.collect()
    print("depth pair %d^3: band %.1f %% of the voxels; chunks written with reach 2: %.1f %%" % (n, 100 * band, 100 * p.needed_fraction()))
