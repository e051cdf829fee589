"""Closed-form synthetic TSDF pairs of SURVEY.md section 8(d) ("sphere-pair"), generated directly on the GPU
with torch (input plumbing, not part of the measured path).

canonical = TSDF of a sphere (circle in 2-D), live = the same sphere translated by (1.5, -1.0, 2.0) voxels and
anisotropically scaled by (1.05, 0.95, 1.0) along (x, y, z); narrow-band half width 10 voxels; values are exactly
+-1 outside the band (the band predicates test == +-1).  For slab runs the pattern repeats every n slices in z.
"""
import torch


def sphere_pair(n, dims=3, device="cuda", z_range=None):
    """returns (canonical, live) float32 tensors of shape (n, n) or (nz, n, n); z_range = (z0, z1) selects the
    global slices [z0, z1) of a volume whose sphere pattern has period n along z"""
    h, r, c = 10.0, 0.3 * n, n / 2.0
    ax = torch.arange(n, dtype=torch.float64, device=device)
    if dims == 2:
        yy, xx = torch.meshgrid(ax, ax, indexing="ij")
        coords = [xx, yy]
    else:
        z0, z1 = (0, n) if z_range is None else z_range
        az = torch.remainder(torch.arange(z0, z1, dtype=torch.float64, device=device), n)
        zz, yy, xx = torch.meshgrid(az, ax, ax, indexing="ij")
        coords = [xx, yy, zz]

    def tsdf(shift, scale):
        sq = None
        for i, q in enumerate(coords):
            t = ((q - (c + shift[i])) / scale[i]) ** 2
            sq = t if sq is None else sq + t
        return torch.clamp((torch.sqrt(sq) - r) / h, -1.0, 1.0).to(torch.float32).contiguous()

    return tsdf((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), tsdf((1.5, -1.0, 2.0), (1.05, 0.95, 1.0))
