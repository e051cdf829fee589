"""The complete N > 1 flow on ONE GPU: two processes (gloo, halos staged through the host) each run the slab engine --
HIP kernels with z-range arguments, per-iteration halo exchange of live and warp, MAX/SUM reduction of the iteration
records, the device-side gate on the reduced record -- and the stitched result must equal the single-process
whole-volume run BIT FOR BIT.  On the 8-GPU node the only difference is the transport (RCCL instead of gloo)."""
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


def _worker(rank, world, port, n, nz, halo, kwargs, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    from levelsetfusion_python_amd.synthetic import sphere_pair
    layout = SlabLayout(nz, rank, world, halo)
    comm = SlabComm(layout)
    sl = layout.local_slice()
    canonical, live = sphere_pair(n, 3, "cuda", (sl.start, sl.stop))
    opt = lsf.SlavchevaOptimizer3d(field_size=n, comm=comm, **kwargs)
    opt.optimize(live, canonical)
    own = layout.owned_local()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), live=live[own].cpu().numpy(),
             warp=opt.warp_field[own].cpu().numpy(), max_warps=np.float32(opt.log.max_warps),
             locations=np.int64(opt.log.max_warp_locations), data=np.float64(opt.log.data_energies),
             smoothing=np.float64(opt.log.smoothing_energies), level_set=np.float64(opt.log.level_set_energies))
    dist.destroy_process_group()


@pytest.mark.parametrize("config", ["killing_fixed", "killing_fixed_halo1", "killing_threshold", "sobolev"])
def test_two_slab_ranks_equal_whole_volume(tmp_path, config):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n, world = 64, 2
    nz = n * world  # two periods of the sphere pattern stacked along z, as in bench.py's weak scaling
    if config == "sobolev":
        halo = 3
        kwargs = dict(compute_method=lsf.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                      sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1),
                      maximum_warp_length_lower_threshold=0.0, max_iterations=4, min_iterations=4, check_interval=3)
    else:
        # one halo slice is enough while every warp update stays below one voxel (stencils reach 1, the re-warp gather
        # floor(|w_z|) + 1): the guard in SlavchevaEngine.optimize raises otherwise
        halo = 1 if config.endswith("halo1") else 2
        kwargs = dict(compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                      smoothing_term_method=lsf.SmoothingTermMethod.KILLING, check_interval=4)
        if config.startswith("killing_fixed"):
            kwargs.update(maximum_warp_length_lower_threshold=0.0, max_iterations=6, min_iterations=6)
        else:  # the gate closes on the REDUCED record: both ranks must stop after the same iteration
            kwargs.update(maximum_warp_length_lower_threshold=0.0319, max_iterations=30, min_iterations=2)
    mp.spawn(_worker, args=(world, _free_port(), n, nz, halo, kwargs, str(tmp_path)), nprocs=world, join=True)
    canonical, live = sphere_pair(n, 3, "cuda", (0, nz))
    ref = lsf.SlavchevaOptimizer3d(field_size=n, **kwargs)
    ref._run_checks = lambda *a: None  # the stacked volume is not a cube
    ref.optimize(live, canonical)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    if config == "killing_threshold":
        assert 2 < len(ref.log.max_warps) < 30
    assert np.array_equal(np.concatenate([p["live"] for p in parts], 0), live.cpu().numpy())
    assert np.array_equal(np.concatenate([p["warp"] for p in parts], 0), ref.warp_field.cpu().numpy())
    for p in parts:
        assert np.array_equal(p["max_warps"], np.float32(ref.log.max_warps))
        want = [np.ravel_multi_index(loc[::-1], (nz, n, n)) for loc in ref.log.max_warp_locations]
        assert list(p["locations"]) == [int(w) for w in want]
        assert np.allclose(p["data"], ref.log.data_energies, rtol=1e-10)
        assert np.allclose(p["smoothing"], ref.log.smoothing_energies, rtol=1e-10)
        assert np.allclose(p["level_set"], ref.log.level_set_energies, rtol=1e-10)


def _report_row(r):
    w, d = r.warp_delta_statistics, r.tsdf_difference_statistics
    return [r.iteration_count, float(r.iteration_limit_reached), w.ratio_above_min_threshold, w.length_max,
            w.length_mean, w.length_standard_deviation, *w.longest_warp_location, d.difference_min, d.difference_max,
            d.difference_mean, d.difference_standard_deviation, *d.biggest_difference_location]


def _hier_worker(rank, world, port, n, nz, halo, kwargs, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    from levelsetfusion_python_amd.synthetic import sphere_pair
    layout = SlabLayout(nz, rank, world, halo)
    sl = layout.local_slice()
    canonical, live = sphere_pair(n, 3, "cuda", (sl.start, sl.stop))
    fused_filter = kwargs.pop("force_fused_filter", False)
    opt = lsf.HierarchicalOptimizer3d(
        comm=SlabComm(layout),
        logging_parameters=lsf.HierarchicalOptimizer3d.LoggingParameters(collect_per_level_convergence_reports=True),
        # every level filters with lsf_convolve_xyz on its owned z-range (default: levels of >= 2^23 voxels)
        engine_options=dict(fused_filter_min_voxels=0) if fused_filter else None,
        **kwargs)
    warp = opt.optimize(canonical, live)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), warp=warp.cpu().numpy(),
             counts=np.int64(opt.get_per_level_iteration_counts()),
             last_max=np.float32([m[-1] for m in opt.get_per_level_maximum_updates()]),
             reports=np.float64([_report_row(r) for r in opt.get_per_level_convergence_reports()]))
    dist.destroy_process_group()


@pytest.mark.parametrize("config", ["tikhonov_fixed", "tikhonov_kernel_fixed", "tikhonov_kernel_fused", "data_threshold",
                                    "linear_halo4", "linear_halo2"])
def test_two_slab_ranks_hierarchical_equal_whole_volume(tmp_path, config):
    """HierarchicalOptimizer3d on two z-slabs (per-level halos, gradient halo exchange, global gate) == whole volume;
    linear_*: ResamplingStrategy.LINEAR (windows of the restriction and the prolongation reach into the halos)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n, world, halo = 64, 2, (2 if config == "linear_halo2" else 4)
    nz = n * world
    kwargs = dict(maximum_chunk_size=4, rate=0.1, tikhonov_strength=0.05, check_interval=3)
    if config.startswith("linear"):
        kwargs.update(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_iteration_count=5,
                      maximum_warp_update_threshold=0.0,
                      resampling_strategy=lsf.HierarchicalOptimizer3d.ResamplingStrategy.LINEAR)
    elif config == "tikhonov_fixed":
        kwargs.update(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_iteration_count=5,
                      maximum_warp_update_threshold=0.0)
    elif config in ("tikhonov_kernel_fixed", "tikhonov_kernel_fused"):
        kwargs.update(tikhonov_term_enabled=True, gradient_kernel_enabled=True,
                      kernel=lsf.generate_1d_sobolev_kernel(7, 0.1), maximum_iteration_count=4,
                      maximum_warp_update_threshold=0.0)
    else:
        kwargs.update(tikhonov_term_enabled=False, gradient_kernel_enabled=False, maximum_iteration_count=40,
                      maximum_warp_update_threshold=0.0)
        # pick a threshold that the coarsest level crosses in mid-run (probe run on the whole volume)
        probe = lsf.HierarchicalOptimizer3d(**kwargs)
        probe.optimize(*sphere_pair(n, 3, "cuda", (0, nz)))
        trajectory = probe.get_per_level_maximum_updates()[0]
        assert trajectory[12] < trajectory[2]
        kwargs["maximum_warp_update_threshold"] = float(0.5 * (trajectory[11] + trajectory[12]))
    worker_kwargs = dict(kwargs, force_fused_filter=True) if config == "tikhonov_kernel_fused" else kwargs
    mp.spawn(_hier_worker, args=(world, _free_port(), n, nz, halo, worker_kwargs, str(tmp_path)), nprocs=world, join=True)
    canonical, live = sphere_pair(n, 3, "cuda", (0, nz))
    ref = lsf.HierarchicalOptimizer3d(
        logging_parameters=lsf.HierarchicalOptimizer3d.LoggingParameters(collect_per_level_convergence_reports=True),
        **kwargs)
    warp = ref.optimize(canonical, live).cpu().numpy()
    want_reports = np.float64([_report_row(r) for r in ref.get_per_level_convergence_reports()])
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    got = np.concatenate([p["warp"] for p in parts], 0)
    per_slice = np.abs(got - warp).reshape(nz, -1).max(1)
    assert np.array_equal(got, warp), [(z, float(d)) for z, d in enumerate(per_slice) if d > 0]
    for p in parts:
        assert list(p["counts"]) == ref.get_per_level_iteration_counts()
        assert np.array_equal(p["last_max"], np.float32([m[-1] for m in ref.get_per_level_maximum_updates()]))
        # per-level convergence reports: the slab ranks combine their statistics into those of the whole volume
        assert p["reports"].shape == want_reports.shape and want_reports.shape[1] == 16
        assert np.allclose(p["reports"], want_reports, rtol=1e-9, atol=1e-12), (p["reports"], want_reports)
    if config == "data_threshold":
        assert any(1 < c < 40 for c in ref.get_per_level_iteration_counts())


def test_halo_copy_kernel_matches_slicing():
    """lsf_halo_copy (the RCCL path's pack / unpack kernel) == the tensor slicing the gloo path uses"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd  # noqa: F401
    from levelsetfusion_python_amd import device as dev
    g = torch.Generator("cuda").manual_seed(7)
    nz, ny, nx, h = 12, 10, 70, 3
    live = torch.randn((nz, ny, nx), device="cuda", generator=g)
    warp = torch.randn((3, nz, ny, nx), device="cuda", generator=g)
    lo = torch.zeros((4, h, ny, nx), device="cuda")
    hi = torch.zeros_like(lo)
    z_lo, z_hi = 3, nz - 3 - h
    dev.halo_copy(live, warp, lo, hi, h, z_lo, z_hi, unpack=False)
    assert torch.equal(lo, torch.cat([live[z_lo:z_lo + h][None], warp[:, z_lo:z_lo + h]]))
    assert torch.equal(hi, torch.cat([live[z_hi:z_hi + h][None], warp[:, z_hi:z_hi + h]]))
    live2, warp2 = live.clone(), warp.clone()
    dev.halo_copy(live2, warp2, hi, lo, h, 0, nz - h, unpack=True)  # swapped messages into the end slices
    assert torch.equal(live2[:h], hi[0]) and torch.equal(warp2[:, :h], hi[1:])
    assert torch.equal(live2[nz - h:], lo[0]) and torch.equal(warp2[:, nz - h:], lo[1:])
    assert torch.equal(live2[h:nz - h], live[h:nz - h])
    # one-sided (end ranks): the missing neighbour's message is None and nothing else is touched
    only = torch.zeros_like(lo)
    dev.halo_copy(live, warp, None, only, h, 0, z_hi, unpack=False)
    assert torch.equal(only, hi)


def test_native_rccl_transport_equals_torch_transport(tmp_path):
    """the library's own RCCL transport (lsf_slab_state_iteration: boundary launches, ncclSend / ncclRecv on the comm
    stream, interior launches in ONE host call; compact band-voxel faces and whole-slice faces; exchange groups) against
    the torch.distributed transport, all over real RCCL on one GPU (a world of one rank that is its own neighbour, see
    slab_loopback_worker.py): bit-identical fields and records, and the middle slab of the whole volume within 2e-5"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    out = os.path.join(str(tmp_path), "loopback.npz")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slab_loopback_worker.py")
    proc = subprocess.run([sys.executable, worker, out, str(_free_port())], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    r = np.load(out)
    assert str(r["used_rccl"]) == "rccl:compact", "native RCCL transport with compact faces was not used: " + \
        str(r["used_rccl"]) + proc.stderr[-1000:]
    assert str(r["used_torch"]) == "torch"
    # compact faces == whole-slice faces == torch.distributed transport, bit for bit
    assert bool(r["live_equal"]) and bool(r["warp_equal"]) and bool(r["max_equal"]) and bool(r["data_close"])
    # ... and == the middle slab of the whole (z-periodic) volume computed by one process
    assert float(r["whole_live_diff"]) <= 2e-5 and float(r["whole_warp_diff"]) <= 2e-5, \
        (float(r["whole_live_diff"]), float(r["whole_warp_diff"]))
    assert float(r["moved"]) > 1e-3  # the optimisation did something
    # the same call on states initialised near the band only (slab exchange groups, compact and whole faces): same bits
    assert str(r["sparse_differs"]) == "", str(r["sparse_differs"])


def test_library_enqueued_slab_call_equals_the_call_made_from_python(tmp_path):
    """lsf_slab_run_begin / _finish (the whole call of a z-slab rank in two foreign calls: cut positions from the counting
    pass, the exchange-group schedule, compact faces, the gather of every rank's records -- round 6) against the same call
    enqueued iteration by iteration from Python and against the torch.distributed transport, over real RCCL in the one-GPU
    loop-back (slab_loopback_worker.py, mode library_run): eight schedules, bit-identical fields, records and gradients"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    out = os.path.join(str(tmp_path), "library_run.npz")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slab_loopback_worker.py")
    proc = subprocess.run([sys.executable, worker, out, str(_free_port()), "library_run"], capture_output=True, text=True,
                          timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    r = np.load(out)
    assert str(r["problems"]) == "", str(r["problems"])
    assert r["taken"].size == 24 and bool(r["taken"].all()), "the library-enqueued slab call was not taken"
