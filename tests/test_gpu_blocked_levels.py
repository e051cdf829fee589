"""2-D pyramid levels advanced EIGHT iterations per launch inside LDS tiles
(lsf_hier_level_run_2d, engine option blocked_levels; round 6: BASELINE config 2 is launch-bound) against the
one-launch-per-iteration path and against the oracle: warp, every iteration's maximum and its location, the last gradient
-- bit for bit (a tile recomputes the rings of voxels around it with the same arithmetic on the same inputs) --, with a
fixed iteration count and with a stop test that fires anywhere inside a launch.
Reference loop: nonrigid_opt/hierarchical/hierarchical_optimizer2d.py:184-225."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


def _run(lsf, canonical, live, blocked, **kw):
    opt = lsf.HierarchicalOptimizer2d(engine_options=dict(blocked_levels=blocked), **kw)
    warp = opt.optimize(canonical, live)
    assert opt.engine.last_call.blocked_levels == (len(opt.engine.level_results) if blocked else 0)
    return opt, warp


def _same(a, b):
    (oa, wa), (ob, wb) = a, b
    assert torch.equal(torch.as_tensor(wa), torch.as_tensor(wb)), "warp"
    assert oa.get_per_level_iteration_counts() == ob.get_per_level_iteration_counts()
    for ma, mb in zip(oa.get_per_level_maximum_updates(), ob.get_per_level_maximum_updates()):
        assert np.array_equal(np.float32(ma), np.float32(mb))
    for ra, rb in zip(oa.engine.level_results, ob.engine.level_results):
        assert list(ra.argmax) == list(rb.argmax)
    assert torch.equal(oa.engine.last_gradient, ob.engine.last_gradient), "the last iteration's gradient"


# iteration counts: a multiple of 8, a remainder, fewer than one launch's worth, one
@pytest.mark.parametrize("n,chunk,iterations", [(512, 4, 100), (256, 8, 16), (128, 4, 13), (64, 8, 3), (64, 4, 1)])
def test_blocked_levels_equal_one_launch_per_iteration(lsf, n, chunk, iterations):
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(n, 2, "cuda")
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=chunk, rate=0.1,
              maximum_iteration_count=iterations, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    a = _run(lsf, canonical, live, True, **kw)
    b = _run(lsf, canonical, live, False, **kw)
    _same(a, b)
    assert float(torch.as_tensor(a[1]).abs().max()) > 1e-3


def test_blocked_levels_on_fields_that_are_not_square(lsf):
    """128 x 512: tiles clipped by the array on one axis only, a level of a single row of tiles"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = (t[192:320].contiguous() for t in sphere_pair(512, 2, "cuda"))
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=20, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    _same(_run(lsf, canonical, live, True, **kw), _run(lsf, canonical, live, False, **kw))


def test_blocked_levels_equal_the_oracle_in_the_divergent_regime(lsf):
    """BASELINE config 2 as SURVEY 8(d) names it (chunk 4, tikhonov_strength 0.2: the recurrence amplifies the highest
    frequency 1.6-fold per iteration) for 24 fixed iterations at 128^2: warp and maxima == oracle"""
    canonical, live = O.sphere_pair(128, d=2)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=24, maximum_warp_update_threshold=0.0, tikhonov_strength=0.2)
    opt, warp = _run(lsf, canonical, live, True, **kw)
    o = O.HierarchicalOracle(**kw)
    want = o.optimize(canonical, live)
    assert np.array_equal(warp, want)
    assert float(np.abs(want).max()) > 1e-3


# with the gradient kernel (the reference's default constructor): the filter's passes and the update run in the launch too;
# an iteration consumes taps / 2 + 1 rings, so a launch holds 4 / 2 / 2 / 1 iterations for 3 / 5 / 7 / 9 taps
@pytest.mark.parametrize("n_taps", [3, 5, 7, 9])
@pytest.mark.parametrize("n,chunk,iterations", [(512, 4, 21), (128, 8, 12), (64, 4, 5)])
def test_filtered_blocked_levels_equal_one_launch_per_iteration(lsf, n_taps, n, chunk, iterations):
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(n, 2, "cuda")
    kernel = lsf.generate_1d_sobolev_kernel(n_taps, 0.1) if n_taps in (3, 7) else \
        np.linspace(-0.1, 0.5, n_taps).astype(np.float64) / 1.7  # (not float32 values: the two-instruction arithmetic)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=True, kernel=kernel, maximum_chunk_size=chunk, rate=0.1,
              maximum_iteration_count=iterations, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    a = _run(lsf, canonical, live, True, **kw)
    b = _run(lsf, canonical, live, False, **kw)
    _same(a, b)
    assert float(torch.as_tensor(a[1]).abs().max()) > 1e-3


def test_filtered_blocked_levels_on_fields_that_are_not_square(lsf):
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = (t[192:320].contiguous() for t in sphere_pair(512, 2, "cuda"))
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=True, kernel=lsf.generate_1d_sobolev_kernel(7, 0.1),
              maximum_chunk_size=4, rate=0.1, maximum_iteration_count=15, maximum_warp_update_threshold=0.0,
              tikhonov_strength=0.05)
    _same(_run(lsf, canonical, live, True, **kw), _run(lsf, canonical, live, False, **kw))


@pytest.mark.parametrize("n_taps", [3, 7])
def test_filtered_blocked_levels_equal_the_oracle(lsf, n_taps):
    """the reference's default configuration (Tikhonov term + gradient kernel, hierarchical_optimizer2d.py:63-73) at 128^2,
    fixed count and with a stop test that fires: warp, counts and maxima == oracle"""
    canonical, live = O.sphere_pair(128, d=2)
    kernel = O.generate_1d_sobolev_kernel(n_taps, 0.1)
    for threshold, iterations in ((0.0, 11), (0.037, 40)):
        kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=True, maximum_chunk_size=8, rate=0.1,
                  maximum_iteration_count=iterations, maximum_warp_update_threshold=threshold, tikhonov_strength=0.05)
        opt, warp = _run(lsf, canonical, live, True, kernel=kernel, **kw)
        o = O.HierarchicalOracle(kernel=kernel, **kw)
        want = o.optimize(canonical, live)
        assert np.array_equal(warp, want)
        assert opt.get_per_level_iteration_counts() == o.per_level_iteration_counts
        for mine, theirs in zip(opt.get_per_level_maximum_updates(), o.per_level_max_updates):
            assert np.array_equal(np.float32(mine), np.float32(theirs))
        if threshold > 0:
            assert min(o.per_level_iteration_counts) < iterations, o.per_level_iteration_counts


@pytest.mark.parametrize("n_taps", [0, 3, 7])
@pytest.mark.parametrize("threshold", [0.0, 0.03])
def test_blocked_levels_without_the_tikhonov_term(lsf, n_taps, threshold):
    """the data term alone (a voxel's update then depends on nothing around it: eight iterations per launch, no recomputed
    rings), and the data term behind the gradient kernel: == the per-iteration path and the oracle"""
    canonical, live = O.sphere_pair(128, d=2)
    kernel = O.generate_1d_sobolev_kernel(n_taps, 0.1) if n_taps else None
    kw = dict(tikhonov_term_enabled=False, gradient_kernel_enabled=bool(n_taps), maximum_chunk_size=8, rate=0.2,
              maximum_iteration_count=19, maximum_warp_update_threshold=threshold)
    opt, warp = _run(lsf, canonical, live, True, kernel=kernel, **kw)
    other, warp_b = _run(lsf, canonical, live, False, kernel=kernel, **kw)
    assert np.array_equal(warp, warp_b)
    assert opt.get_per_level_iteration_counts() == other.get_per_level_iteration_counts()
    o = O.HierarchicalOracle(kernel=kernel, **kw)
    want = o.optimize(canonical, live)
    assert np.array_equal(warp, want) and float(np.abs(want).max()) > 1e-3
    assert opt.get_per_level_iteration_counts() == o.per_level_iteration_counts
    for mine, theirs in zip(opt.get_per_level_maximum_updates(), o.per_level_max_updates):
        assert np.array_equal(np.float32(mine), np.float32(theirs))


def test_energy_printouts_keep_the_per_iteration_path(lsf):
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(64, 2, "cuda")
    opt = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8,
                                      rate=0.1, maximum_iteration_count=10, tikhonov_strength=0.05,
                                      maximum_warp_update_threshold=0.0,
                                      verbosity_parameters=lsf.HierarchicalOptimizer2d.VerbosityParameters(
                                          print_iteration_data_energy=True))
    opt.optimize(canonical, live)
    assert opt.engine.last_call.blocked_levels == 0


# where in its launch of eight the level converges: its first iteration (a repeated launch of one), the last of the first
# launch, the first / the middle / the last of the second, the middle of the short third launch (20 - 16 = 4 iterations),
# the last iteration allowed, never
@pytest.mark.parametrize("stop_at", [0, 7, 8, 10, 12, 15, 17, 19, None])
def test_stop_test_inside_a_launch(lsf, stop_at):
    """four levels (16^2 ... 128^2), 20 iterations at most; the threshold is put just above the maximum of iteration
    `stop_at` of the coarsest level of an unterminated run, so the reference leaves that level behind that iteration (the
    finer levels end where their own maxima say): counts, maxima, warp and last gradient are the per-iteration path's and
    the oracle's"""
    n = 128
    canonical, live = O.sphere_pair(n, d=2)
    base = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8, rate=0.1,
                maximum_iteration_count=20, tikhonov_strength=0.05)
    free = O.HierarchicalOracle(maximum_warp_update_threshold=0.0, **base)
    free.optimize(canonical, live)
    maxima = np.float32(free.per_level_max_updates[0])
    assert len(maxima) == 20
    threshold = 1e-12 if stop_at is None else float(np.nextafter(maxima[stop_at], np.float32(np.inf)))
    assert stop_at is None or np.all(maxima[:stop_at] >= np.float32(threshold))  # ... and no earlier iteration is below it
    kw = dict(base, maximum_warp_update_threshold=threshold)
    a = _run(lsf, canonical, live, True, **kw)
    b = _run(lsf, canonical, live, False, **kw)
    _same(a, b)
    assert a[0].get_per_level_iteration_counts()[0] == (20 if stop_at is None else stop_at + 1)
    o = O.HierarchicalOracle(**kw)
    want = o.optimize(canonical, live)
    assert np.array_equal(a[1], want)
    assert a[0].get_per_level_iteration_counts() == o.per_level_iteration_counts
    for mine, theirs in zip(a[0].get_per_level_maximum_updates(), o.per_level_max_updates):
        assert np.array_equal(np.float32(mine), np.float32(theirs))


@pytest.mark.parametrize("n,chunk,threshold", [(128, 8, 0.033), (256, 8, 0.044), (512, 4, 0.0875)])
def test_threshold_terminated_pyramids(lsf, n, chunk, threshold):
    """several levels, each ending where its own maxima say (the coarse ones early, the finest may run out)"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(n, 2, "cuda")
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=chunk, rate=0.1,
              maximum_iteration_count=60, maximum_warp_update_threshold=threshold, tikhonov_strength=0.05)
    a = _run(lsf, canonical, live, True, **kw)
    b = _run(lsf, canonical, live, False, **kw)
    _same(a, b)
    counts = a[0].get_per_level_iteration_counts()
    assert min(counts) < 60 and len(set(counts)) >= 3, counts


def test_threshold_terminated_levels_on_fields_that_are_not_square(lsf):
    """128 x 512 with a stop test: tiles clipped on one axis, levels that end inside a launch"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = (t[192:320].contiguous() for t in sphere_pair(512, 2, "cuda"))
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=45, maximum_warp_update_threshold=0.0875, tikhonov_strength=0.05)
    a = _run(lsf, canonical, live, True, **kw)
    _same(a, _run(lsf, canonical, live, False, **kw))
    counts = a[0].get_per_level_iteration_counts()
    assert any(c % 8 not in (0, 45 % 8) for c in counts) or min(counts) < 45, counts


@pytest.mark.parametrize("threshold", [0.0, 0.033])
def test_per_level_convergence_reports_of_blocked_levels(lsf, threshold):
    """the reports are made at the end of the call, from what every level left behind: equal to the per-iteration path's"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(128, 2, "cuda")
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8, rate=0.1,
              maximum_iteration_count=30, maximum_warp_update_threshold=threshold, tikhonov_strength=0.05,
              logging_parameters=lsf.HierarchicalOptimizer2d.LoggingParameters(collect_per_level_convergence_reports=True))
    a, b = _run(lsf, canonical, live, True, **kw), _run(lsf, canonical, live, False, **kw)
    _same(a, b)
    ra, rb = a[0].get_per_level_convergence_reports(), b[0].get_per_level_convergence_reports()
    assert len(ra) == len(rb) == 4 and all(x == y for x, y in zip(ra, rb)), (ra, rb)
    assert [r.iteration_count for r in ra] == a[0].get_per_level_iteration_counts()
    if threshold > 0:
        assert not ra[0].iteration_limit_reached and ra[-1].iteration_limit_reached


def test_entry_point_refuses_what_it_does_not_implement(lsf):
    import ctypes
    from levelsetfusion_python_amd import _lib
    g2, g3 = _lib.Grid(2, 1, 64, 64, 0, 1, 0, 0), _lib.Grid(3, 8, 64, 64, 0, 8, 0, 0)
    ok = _lib.HierParams(1.0, 0.05, 0.1, 1, 1, 0)
    call = _lib.lib.lsf_hier_level_run_2d
    one, two, three, four, five = (ctypes.c_void_p(k * 4096) for k in (1, 2, 3, 4, 5))
    assert call(one, one, two, three, four, five, ctypes.byref(g3), ctypes.byref(ok), None, 0, one, 4, 8, 0.0, None) == -2  # 3-D
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(_lib.HierParams(1.0, 0.05, 0.1, 1, 0, 0)), None, 0,
                one, 4, 8, 0.0, None) == -2  # the update left to a filter that is not there
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(_lib.HierParams(1.0, 0.05, 0.1, 1, 1, 1)), None, 0,
                one, 4, 8, 0.0, None) == -2  # energies
    assert call(one, one, two, two, four, five, ctypes.byref(g2), ctypes.byref(ok), None, 0, one, 4, 8, 0.0, None) == -1    # same buffer twice
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(ok), None, 0, one, 4, 9, 0.0, None) == -1  # K > 8
