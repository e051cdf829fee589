"""z-slab runs beyond two ranks and beyond sub-voxel updates, on ONE GPU (gloo, halos staged through the host; at most 4
worker processes on the card): three and four slabs -- middle ranks with two neighbours --, and a pair whose warp
updates are several voxels long (the reference's orthographic pair, embedded in 3-D so that the large motion crosses the
slab faces): the engine discards the attempt and re-runs it on a wider internal slab (SlavchevaEngine.optimize).  The
stitched result equals the single-process whole-volume run BIT FOR BIT in every case."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _volume(kind, n, nz):
    """(canonical, live) float32 numpy [nz, n, n]"""
    sys.path.insert(0, ROOT)
    if kind == "sphere":
        from oracle import lsf_oracle as O
        c, l = O.sphere_pair(n, d=3, nz=nz, z_total=nz)
        return c, l
    # the reference's orthographic 2-D pair (tests/golden: generate_initial_orthographic_2d_tsdf_fields, 64 x 64) laid
    # into the (z, x) plane and swept along y with a slow shear: default KillingFusion weights move it by several voxels
    # per iteration, along z -- across the slab faces -- as much as along x
    S = np.load(os.path.join(ROOT, "tests", "golden", "ref_slavcheva.npz"), allow_pickle=False)
    live2, canon2 = S["ortho64.live"], S["ortho64.canonical"]
    assert live2.shape == (n, n) and nz % n == 0
    out = []
    for f in (canon2, live2):
        vol = np.empty((nz, n, n), np.float32)
        for y in range(n):
            vol[:, y, :] = np.tile(np.roll(f, y // 16, axis=1), (nz // n, 1))  # the pattern repeats along z
        out.append(vol)
    return out[0], out[1]


def _worker(rank, world, port, kind, n, nz, halo, kwargs, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    layout = SlabLayout(nz, rank, world, halo)
    comm = SlabComm(layout)
    sl = layout.local_slice()
    canonical, live = (torch.from_numpy(np.ascontiguousarray(v[sl])).cuda() for v in _volume(kind, n, nz))
    opt = lsf.SlavchevaOptimizer3d(field_size=n, comm=comm, **kwargs)
    opt.optimize(live, canonical)
    own = layout.owned_local()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), live=live[own].cpu().numpy(),
             warp=opt.warp_field[own].cpu().numpy(), max_warps=np.float32(opt.log.max_warps),
             data=np.float64(opt.log.data_energies))
    dist.destroy_process_group()


CASES = {
    # name: (world, volume, halo, fixed iteration count or None for a threshold-terminated run, sobolev)
    "three_slabs_groups": (3, "sphere", 2, 6, False),
    "four_slabs_groups": (4, "sphere", 4, 9, False),
    "four_slabs_threshold": (4, "sphere", 1, None, False),
    "three_slabs_sobolev_lists": (3, "sphere", 3, 4, True),
    "two_slabs_large_updates": (2, "ortho", 2, 5, False),
    "four_slabs_large_updates": (4, "ortho", 2, 4, False),
    # gated run: an update longer than the halo is followed by the stop test firing INSIDE the same check_interval batch
    # (the reach check has to look at a batch's executed iterations before the loop leaves)
    "two_slabs_large_update_then_stop": (2, "ortho", 2, "upper", False),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_slab_ranks_equal_whole_volume(tmp_path, case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    import levelsetfusion_python_amd as lsf
    world, kind, halo, fixed, sobolev = CASES[case]
    n = 64
    nz = 96 if (world == 3 and kind == "sphere") else (128 if (kind == "sphere" or world == 4) else 64)
    if sobolev:
        kwargs = dict(compute_method=lsf.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                      sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1), check_interval=3)
    else:
        kwargs = dict(compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                      smoothing_term_method=lsf.SmoothingTermMethod.KILLING, check_interval=4)
    if fixed == "upper":
        # the whole volume, ungated, tells where the first update longer than the 2-slice halo happens; the upper
        # threshold is put between the halo and that update, so the gated run executes it and then stops
        probe = lsf.SlavchevaOptimizer3d(field_size=n, maximum_warp_length_lower_threshold=0.0, max_iterations=5,
                                         min_iterations=5, **kwargs)
        probe._run_checks = lambda *a: None
        pc, pl = (torch.from_numpy(v).cuda() for v in _volume(kind, n, nz))
        probe.optimize(pl, pc)
        k = next(i for i, m in enumerate(probe.log.max_warps) if m > 2.0)
        kwargs.update(maximum_warp_length_lower_threshold=0.0,
                      maximum_warp_length_upper_threshold=0.5 * (2.0 + probe.log.max_warps[k]), max_iterations=8,
                      min_iterations=1, check_interval=8)
    elif fixed is not None:
        kwargs.update(maximum_warp_length_lower_threshold=0.0, max_iterations=fixed, min_iterations=fixed)
    else:
        kwargs.update(maximum_warp_length_lower_threshold=0.0319, max_iterations=30, min_iterations=2)
    mp.spawn(_worker, args=(world, _free_port(), kind, n, nz, halo, kwargs, str(tmp_path)), nprocs=world, join=True)
    canonical, live = (torch.from_numpy(v).cuda() for v in _volume(kind, n, nz))
    ref = lsf.SlavchevaOptimizer3d(field_size=n, **kwargs)
    ref._run_checks = lambda *a: None  # the stacked volume is not a cube
    ref.optimize(live, canonical)
    if kind == "ortho":
        assert max(ref.log.max_warps) > 2.0, "this pair is meant to move by several voxels per iteration"
    if fixed is None:
        assert 2 < len(ref.log.max_warps) < 30
    if fixed == "upper":
        assert len(ref.log.max_warps) == k + 1 < 8 and ref.log.max_warps[-1] > 2.0
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([p["live"] for p in parts], 0), live.cpu().numpy())
    assert np.array_equal(np.concatenate([p["warp"] for p in parts], 0), ref.warp_field.cpu().numpy())
    for p in parts:
        assert np.array_equal(p["max_warps"], np.float32(ref.log.max_warps))
        assert np.allclose(p["data"], ref.log.data_energies, rtol=1e-10)
