"""Closed-form synthetic TSDF pairs of SURVEY.md section 8(d) ("sphere-pair"), generated directly on the GPU
with torch (input plumbing, not part of the measured path).

canonical = TSDF of a sphere (circle in 2-D), live = the same sphere translated by (1.5, -1.0, 2.0) voxels and
anisotropically scaled by (1.05, 0.95, 1.0) along (x, y, z); narrow-band half width 10 voxels; values are exactly
+-1 outside the band (the band predicates test == +-1).  For slab runs the pattern repeats every n slices in z.
"""
import torch


def sphere_pair(n, dims=3, device="cuda", z_range=None, z_shift=0):
    """returns (canonical, live) float32 tensors of shape (n, n) or (nz, n, n); z_range = (z0, z1) selects the
    global slices [z0, z1) of a volume whose sphere pattern has period n along z; z_shift moves the pattern by that
    many slices: with z_shift = n / 2 the spheres are centred ON the faces z = 0, n, 2n, ... of a stack of n-slice slabs,
    so that every slab face cuts a narrow band through its equator (an annulus of ~15 % of the face) and a z-slab run's
    halo exchange carries data -- bench.py's weak-scaling input"""
    h, r, c = 10.0, 0.3 * n, n / 2.0
    ax = torch.arange(n, dtype=torch.float64, device=device)
    if dims == 2:
        yy, xx = torch.meshgrid(ax, ax, indexing="ij")
        coords = [xx, yy]
    else:
        z0, z1 = (0, n) if z_range is None else z_range
        az = torch.remainder(torch.arange(z0, z1, dtype=torch.float64, device=device) + float(z_shift), n)
        zz, yy, xx = torch.meshgrid(az, ax, ax, indexing="ij")
        coords = [xx, yy, zz]

    def tsdf(shift, scale):
        sq = None
        for i, q in enumerate(coords):
            t = ((q - (c + shift[i])) / scale[i]) ** 2
            sq = t if sq is None else sq + t
        return torch.clamp((torch.sqrt(sq) - r) / h, -1.0, 1.0).to(torch.float32).contiguous()

    return tsdf((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), tsdf((1.5, -1.0, 2.0), (1.05, 0.95, 1.0))


FRAME_STEP = (1.0, -0.5, 1.0)  # voxels per frame along (x, y, z): 1.5 voxels of motion from frame to frame


def sphere_frame(n, k, device="cuda"):
    """frame k of the synthetic multi-frame sequence (BASELINE config 5): the TSDF of the sphere of sphere_pair, its
    centre moved by k * FRAME_STEP voxels; consecutive frames form the (canonical, live) pairs
    (experiment/multiframe_experiment.py:186-233: live_frame_index = canonical_frame_index + 1)"""
    h, r, c = 10.0, 0.3 * n, n / 2.0
    ax = torch.arange(n, dtype=torch.float64, device=device)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    sq = None
    for q, step in zip((xx, yy, zz), FRAME_STEP):
        t = (q - (c + k * step)) ** 2
        sq = t if sq is None else sq + t
    return torch.clamp((torch.sqrt(sq) - r) / h, -1.0, 1.0).to(torch.float32).contiguous()


def depth_image(shift_px=0.0, nearer_m=0.0, width=640, height=480):
    """SURVEY 8(d) "depth->TSDF" input: a tilted plane at ~1 m with a sinusoidal bump, uint16 millimetres (numpy)"""
    import numpy as np
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    bump = 0.06 * np.exp(-(((u - 320.0 - shift_px) / 90.0) ** 2 + ((v - 240.0) / 70.0) ** 2)) \
        * (1.0 + 0.3 * np.sin(u / 17.0) * np.cos(v / 23.0))
    depth_m = 1.0 + 0.0002 * (u - 320.0) - bump - nearer_m
    return np.round(depth_m * 1000.0).astype(np.uint16)


def depth_pair(n, device="cuda"):
    """(canonical, live) n^3 TSDF volumes generated ON THE GPU from two synthetic depth frames (the second with the bump
    shifted 2 px and 8 mm nearer): K = [[700, 0, 320], [0, 700, 240], [0, 0, 1]], depth unit 0.001, the surface at 1 m
    crosses the middle of the volume; 4 mm voxels up to n = 256, scaled down beyond so that the scene still fills it"""
    import numpy as np
    from .tsdf import generation as gen
    K = np.array([[700.0, 0.0, 320.0], [0.0, 700.0, 240.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    cam = gen.DepthCamera(intrinsic_matrix=K, depth_unit_ratio=0.001)
    voxel = 0.004 * min(1.0, 256.0 / n)
    surface = int(round(1.0 / voxel))
    offset = np.array([-n // 2, -n // 2, surface - n // 2])
    frames = (depth_image(), depth_image(shift_px=2.0, nearer_m=0.008))
    with torch.cuda.device(torch.device(device)):
        return tuple(gen.generate_3d_tsdf_field_from_depth_image(d, cam, field_size=n, voxel_size=voxel,
                                                                 array_offset=offset, as_tensor=True) for d in frames)
