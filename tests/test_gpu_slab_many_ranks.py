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
    if kind == "island":
        # ONE sphere pair in the middle third of the stack, free space (+1) everywhere else: the end slabs hold no band
        # voxel at all -- what the end ranks of bench.py --scaling strong see at N = 8 (the 256^3 sphere spans z 41..217)
        from oracle import lsf_oracle as O
        assert nz == 3 * n
        c1, l1 = O.sphere_pair(n, d=3)
        out = []
        for f in (c1, l1):
            vol = np.ones((nz, n, n), np.float32)
            vol[n:2 * n] = f
            out.append(vol)
        return out[0], out[1]
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
    kwargs = dict(kwargs)
    env = kwargs.pop("_env", {})
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    if "LSF_SPARSE_MIN_VOXELS" in env:
        kwargs["engine_options"] = dict(sparse_min_voxels=int(env["LSF_SPARSE_MIN_VOXELS"]))
    layout = SlabLayout(nz, rank, world, halo)
    comm = SlabComm(layout)
    sl = layout.local_slice()
    canonical, live = (torch.from_numpy(np.ascontiguousarray(v[sl])).cuda() for v in _volume(kind, n, nz))
    opt = lsf.SlavchevaOptimizer3d(field_size=n, comm=comm, **kwargs)
    opt.optimize(live, canonical)
    if env.get("LSF_SPARSE_MIN_VOXELS") == "0" and max(opt.log.max_warps) < 1.0:
        assert opt.engine.last_call.sparse_states, "this case is meant to run on states initialised near the band only"
    own = layout.owned_local()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), live=live[own].cpu().numpy(),
             warp=opt.warp_field[own].cpu().numpy(), max_warps=np.float32(opt.log.max_warps),
             data=np.float64(opt.log.data_energies))
    dist.destroy_process_group()


CASES = {
    # name: (world, volume, halo, fixed iteration count or None for a threshold-terminated run, sobolev)
    "three_slabs_groups": (3, "sphere", 2, 6, False),
    "three_slabs_empty_end_ranks": (3, "island", 2, 6, False),
    # the ping-pong states initialised near the band only (engine option sparse_reach; volumes of 2^21 voxels and more by
    # default, every volume here): whole faces travel on this transport, i.e. also voxels a rank never initialised
    "three_slabs_groups_sparse_states": (3, "sphere", 2, 6, "sparse"),
    "two_slabs_large_updates_sparse_states": (2, "ortho", 2, 5, "sparse"),
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
    env = {"LSF_SPARSE_MIN_VOXELS": "0"} if sobolev == "sparse" else {}
    sobolev = sobolev is True
    n = 64
    nz = 96 if (world == 3 and kind == "sphere") else (128 if (kind == "sphere" or world == 4) else 64)
    if kind == "island":
        nz = 3 * n
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
    mp.spawn(_worker, args=(world, _free_port(), kind, n, nz, halo, dict(kwargs, _env=env), str(tmp_path)),
             nprocs=world, join=True)
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


# ---- the hierarchical optimizer on z-slabs when the cumulative warp outgrows the halo ------------------------------------
def _shifted_spheres(n, nz):
    """(canonical, live) float32 numpy [nz, n, n]: a sphere and the same sphere moved by (1, -1, 5) voxels along
    (x, y, z) -- the optimizer has to build up a warp of several slices along z, across the slab faces"""
    z, y, x = np.meshgrid(np.arange(nz, dtype=np.float64), np.arange(n, dtype=np.float64),
                          np.arange(n, dtype=np.float64), indexing="ij")

    def tsdf(cx, cy, cz):
        d = np.sqrt((x - cx) ** 2 + (y - cy) ** 2 + (z - cz) ** 2)
        return np.clip((d - 0.28 * n) / 10.0, -1.0, 1.0).astype(np.float32)
    return tsdf(n / 2.0, n / 2.0, nz / 2.0), tsdf(n / 2.0 + 1.0, n / 2.0 - 1.0, nz / 2.0 + 5.0)


def _hier_worker(rank, world, port, n, nz, halo, kwargs, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    layout = SlabLayout(nz, rank, world, halo)
    sl = layout.local_slice()
    canonical, live = (torch.from_numpy(np.ascontiguousarray(v[sl])).cuda() for v in _shifted_spheres(n, nz))
    opt = lsf.HierarchicalOptimizer3d(
        comm=SlabComm(layout),
        logging_parameters=lsf.HierarchicalOptimizer3d.LoggingParameters(collect_per_level_convergence_reports=True),
        **kwargs)
    warp = opt.optimize(canonical, live)
    reports = opt.get_per_level_convergence_reports()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), warp=warp.cpu().numpy(),
             counts=np.int64(opt.get_per_level_iteration_counts()),
             last_max=np.float32([m[-1] for m in opt.get_per_level_maximum_updates()]),
             replicated=np.int64(opt.engine.replicated_levels),
             diff_max=np.float64([r.tsdf_difference_statistics.difference_max for r in reports]),
             diff_mean=np.float64([r.tsdf_difference_statistics.difference_mean for r in reports]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,config", [(2, "data"), (4, "data"), (2, "tikhonov_kernel")])
def test_hierarchical_slabs_follow_warps_past_the_halo(tmp_path, world, config):
    """the reference never aborts on a long warp (hierarchical_optimizer2d.py:169-171 tests the update threshold only), and
    neither does a z-slab run: when the cumulative |w_z| reaches the halo of the static packed field, all ranks together
    restart the level on a copy of the whole level's packed field -- bit-equal to the whole-volume run"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    import levelsetfusion_python_amd as lsf
    n, nz = 64, 128
    if config == "data":
        halo = 2
        kwargs = dict(maximum_chunk_size=8, rate=1.0, tikhonov_term_enabled=False, gradient_kernel_enabled=False,
                      maximum_iteration_count=60, maximum_warp_update_threshold=0.0, check_interval=8)
    else:
        halo = 4
        kwargs = dict(maximum_chunk_size=8, rate=2.0, tikhonov_term_enabled=True, tikhonov_strength=0.05,
                      gradient_kernel_enabled=True, kernel=lsf.generate_1d_sobolev_kernel(7, 0.1),
                      maximum_iteration_count=60, maximum_warp_update_threshold=0.0, check_interval=8)
    mp.spawn(_hier_worker, args=(world, _free_port(), n, nz, halo, kwargs, str(tmp_path)), nprocs=world, join=True)
    canonical, live = (torch.from_numpy(v).cuda() for v in _shifted_spheres(n, nz))
    ref = lsf.HierarchicalOptimizer3d(
        logging_parameters=lsf.HierarchicalOptimizer3d.LoggingParameters(collect_per_level_convergence_reports=True),
        **kwargs)
    warp = ref.optimize(canonical, live).cpu().numpy()
    assert float(np.abs(warp[..., 2]).max()) > halo, "this pair is meant to need a warp longer than the halo"
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert all(int(p["replicated"]) >= 1 for p in parts), "the run was meant to outgrow its halo"
    got = np.concatenate([p["warp"] for p in parts], 0)
    assert np.array_equal(got, warp)
    reports = ref.get_per_level_convergence_reports()
    for p in parts:
        assert list(p["counts"]) == ref.get_per_level_iteration_counts()
        assert np.array_equal(p["last_max"], np.float32([m[-1] for m in ref.get_per_level_maximum_updates()]))
        assert np.allclose(p["diff_max"], [r.tsdf_difference_statistics.difference_max for r in reports], rtol=1e-9)
        assert np.allclose(p["diff_mean"], [r.tsdf_difference_statistics.difference_mean for r in reports], rtol=1e-9)
