"""Parity AT the bench's own configuration (BASELINE config 4: 3-D KillingFusion -- Killing + level-set, DIRECT,
rate 0.1, weights 1 / 0.2 / 0.2, lambda 0.1 -- FIFTY fixed iterations), not a 3-iteration excerpt of it.

`warp_field_advanced` zeroes a voxel's warp where 1 - |v| < 1e-6 (field_warping.py:138-141): one ulp of difference in
the re-warped value becomes a warp-sized difference, and after 50 iterations it would have spread.  So the long runs
are compared bit for bit: fields, every iteration's maximum and arg-max; energies (float64 sums of float32 terms whose
order varies with the launch) to 1e-9.  Reference being restated through the oracle:
nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330 (3-D rules: DESIGN.md section 3, parity unpinned by any
reference artefact -- the reference's 3-D optimizer is absent C++)."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu

BENCH = dict(level_set_term_enabled=True, gradient_descent_rate=0.1, data_term_weight=1.0, smoothing_term_weight=0.2,
             isomorphic_enforcement_factor=0.1, level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.0)


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


def exact(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max()) == 0.0


def run_pair(lsf, canonical, live0, iterations, **extra):
    n = canonical.shape[0]
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING, max_iterations=iterations,
                                   min_iterations=iterations, check_interval=iterations, **BENCH, **extra)
    live = live0.copy()
    opt.optimize(live, canonical)
    return opt, live


def run_oracle(canonical, live0, iterations):
    o = O.SlavchevaOracle(compute_method=O.DIRECT, smoothing_term_method=O.KILLING, max_iterations=iterations,
                          min_iterations=iterations, **BENCH)
    live = live0.copy()
    o.optimize(live, canonical)
    return o, live


def assert_same_run(opt, live, o, live_ref):
    assert len(opt.log.max_warps) == o.iteration_count
    assert exact(live, live_ref), "warped live field"
    assert exact(opt.warp_field, o.warp_field), "warp field"
    assert exact(opt.gradient_field, o.gradient_field), "gradient field"
    # every iteration's maximum warp length (float32 value) and its location, first maximum in C order
    assert np.array_equal(np.asarray(opt.log.max_warps, dtype=np.float32),
                          np.asarray(o.log["max_warps"], dtype=np.float32))
    assert [tuple(int(v) for v in at) for at in opt.log.max_warp_locations] == \
        [tuple(at[::-1]) for at in o.log["max_warp_locations"]]
    for mine, theirs in ((opt.log.data_energies, o.log["data_energies"]),
                         (opt.log.smoothing_energies, o.log["smoothing_energies"]),
                         (opt.log.level_set_energies, o.log["level_set_energies"])):
        assert np.allclose(mine, theirs, rtol=1e-9, atol=1e-12)


_ORACLE_RUNS = {}


def oracle_run(kind, n, iterations):
    """the oracle's run of a case, made once per module (128^3 x 50 iterations takes 25 s)"""
    key = (kind, n, iterations)
    if key not in _ORACLE_RUNS:
        if kind == "sphere":
            canonical, live0 = O.sphere_pair(n, d=3)
        else:
            from levelsetfusion_python_amd.synthetic import depth_pair
            canonical, live0 = (t.cpu().numpy() for t in depth_pair(n))
        _ORACLE_RUNS[key] = (canonical, live0) + run_oracle(canonical, live0, iterations)
    return _ORACLE_RUNS[key]


# box_walk None: the engine's own choice at these sizes, the LIST walk (slavcheva_state_kernel<..., LIST>); True: the
# 512^3 dominant kernel, slavcheva_state_box_kernel (4 x 4 x 4 boxes staged through LDS), held DIRECTLY against the oracle
@pytest.mark.parametrize("box_walk", [None, True])
@pytest.mark.parametrize("n", [64, 128])
def test_sphere_pair_50_iterations_equal_the_oracle(lsf, n, box_walk):
    """the bench's generator and configuration at sizes the oracle finishes in seconds (64^3: 3 s, 128^3: 25 s)"""
    canonical, live0, o, live_ref = oracle_run("sphere", n, 50)
    opt, live = run_pair(lsf, canonical, live0, 50, engine_options=dict(box_walk=box_walk))
    assert opt.engine.last_call.box_walk == bool(box_walk) and opt.engine.last_call.library_run
    assert_same_run(opt, live, o, live_ref)
    assert max(opt.log.max_warps) < 1.0 and min(opt.log.max_warps) > 0.0  # the run moves, and stays in the taps' reach


@pytest.mark.parametrize("box_walk", [None, True])
def test_depth_pair_20_iterations_equal_the_oracle(lsf, box_walk):
    """SURVEY 8(d)'s other generator: two synthetic depth frames -> TSDF volumes (on the GPU) -> 20 iterations.  Its first
    update is several voxels long: the box walk's re-warp cell leaves the staged shell there (its global-gather branch)"""
    canonical, live0, o, live_ref = oracle_run("depth", 64, 20)
    opt, live = run_pair(lsf, canonical, live0, 20, engine_options=dict(box_walk=box_walk))
    assert opt.engine.last_call.box_walk == bool(box_walk)
    assert_same_run(opt, live, o, live_ref)
    assert float(np.abs(live - live0).max()) > 0.0


@pytest.mark.parametrize("n,iterations", [(256, 50), (512, 20)])
def test_full_size_list_walk_equals_dense_walk(lsf, n, iterations):
    """BASELINE config 4 at full size (and the 512^3 of config 5), 50 (20) iterations: the band-list walk (what the bench
    times: INTERIOR lists, CU-sized workgroups, the taps-in-registers re-warp; streaming stores at 512^3) against the SAME
    kernel walking every voxel -- fields and the records of every iteration bit for bit.  Size-independent property; the
    oracle would need 3.5 minutes at 256^3."""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live0 = sphere_pair(n, 3, "cuda")
    runs = []
    for use_list in (True, False):
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING, max_iterations=iterations,
                                       min_iterations=iterations, check_interval=iterations,
                                       engine_options=dict(use_band_list=use_list), **BENCH)
        live = live0.clone()
        opt.optimize(live, canonical)
        runs.append((opt, live))
    (a, live_a), (b, live_b) = runs
    assert torch.equal(live_a, live_b)
    assert torch.equal(torch.as_tensor(a.warp_field), torch.as_tensor(b.warp_field))
    assert np.array_equal(np.asarray(a.log.max_warps, dtype=np.float32), np.asarray(b.log.max_warps, dtype=np.float32))
    assert list(a.log.max_warp_locations) == list(b.log.max_warp_locations)
    for k in ("data_energies", "smoothing_energies", "level_set_energies"):
        assert np.allclose(getattr(a.log, k), getattr(b.log, k), rtol=1e-9, atol=1e-12)
    ra, rb = a.get_convergence_report(), b.get_convergence_report()
    assert ra.warp_delta_statistics.longest_warp_location == rb.warp_delta_statistics.longest_warp_location
    assert ra.warp_delta_statistics.length_max == rb.warp_delta_statistics.length_max
    assert ra.tsdf_difference_statistics.difference_max == rb.tsdf_difference_statistics.difference_max


def test_repeated_calls_reuse_the_first_calls_device_memory(lsf):
    """an optimizer lets go of the previous call's ping-pong states before it allocates the next call's: from the second
    call on nothing is allocated on the device any more (512^3: the first three calls took 120 / 131 / 70 ms against
    9 ms before that), and the results of every call are the same"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n = 128
    canonical, live0 = sphere_pair(n, 3, "cuda")
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING, max_iterations=5,
                                   min_iterations=5, check_interval=5, **BENCH)
    live = torch.empty_like(live0)
    results, reserved, allocs = [], [], []
    for _ in range(5):
        live.copy_(live0)
        opt.optimize(live, canonical)
        torch.cuda.synchronize()
        results.append(live.clone())
        reserved.append(torch.cuda.memory_reserved())
        allocs.append(torch.cuda.memory_stats()["num_device_alloc"])
    assert all(torch.equal(r, results[0]) for r in results[1:])
    assert reserved[1:] == [reserved[1]] * 4 and allocs[1:] == [allocs[1]] * 4
    g = opt.gradient_field  # still there after the call (recomputed from the states the optimizer keeps until the next)
    assert g is not None and g.shape == (n, n, n, 3)
