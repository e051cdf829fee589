"""Parity where the reference's own defaults take it: out of the well-behaved regime.

* `HierarchicalOptimizer2d()` with its CLASS DEFAULTS (hierarchical_optimizer2d.py:62-77: tikhonov_strength 0.2, rate 0.1,
  100 iterations, threshold 0.001, chunk 8; SURVEY 8(d) config 2 names strength 0.2 as well).  The Tikhonov recurrence
  g <- a d - s Laplace(g_prev) (:198-212) amplifies the highest spatial frequency by 4 D s = 1.6 per iteration in 2-D: the
  coarsest level runs into the iteration limit with updates of 1e15 voxels, the finer levels inherit that warp, gather
  nothing but out-of-bounds taps (value 1 / replacement 0), see a zero gradient and stop after ONE iteration.  With 400
  iterations the coarsest level overflows float32 and the warp is NaN.  (The reference's Python `warp_field` raises on a
  NaN position -- `math.floor(nan)`, utils/sampling.py:139-175 --, so its own behaviour there is an exception; the oracle's
  vectorised restatement carries the NaNs through, and that is what the kernels are held to: no fault, the same
  iteration counts, the same maxima up to the first non-finite one, the same finite mask and the same finite values.)
* `SlavchevaOptimizer2d` leaving its loop through the UPPER warp threshold (slavcheva_optimizer2d.py:360-362: the loop
  runs while lo < max_warp < hi): the iteration that crosses `maximum_warp_length_upper_threshold` is the last one.
"""
import warnings

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


def _oracle_run(canonical, live, **kw):
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")  # overflow / invalid value: the point of the exercise
        o = O.HierarchicalOracle(**kw)
        return o, o.optimize(canonical, live)


def _compare_levels(opt, o):
    assert opt.get_per_level_iteration_counts() == o.per_level_iteration_counts
    first_bad = None
    for level, (a, b) in enumerate(zip(opt.get_per_level_maximum_updates(), o.per_level_max_updates)):
        a, b = np.float32(a), np.float32(b)
        assert a.shape == b.shape
        finite = np.isfinite(b)
        # every maximum up to the first non-finite one is THE SAME float32; from there on both sides are non-finite
        # together (an infinity stays the same infinity, NaN payloads are not compared)
        k = int(np.argmin(finite)) if not finite.all() else len(b)
        assert np.array_equal(a[:k], b[:k]), "level %d" % level
        assert np.array_equal(np.isfinite(a), finite), "level %d" % level
        assert np.array_equal(np.isnan(a), np.isnan(b)), "level %d" % level
        assert np.array_equal(a[np.isinf(b)], b[np.isinf(b)]), "level %d" % level
        if k < len(b) and first_bad is None:
            first_bad = (level, k)
    return first_bad


def _compare_warps(warp, ref):
    warp, ref = np.asarray(warp), np.asarray(ref)
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(warp), finite)
    assert np.array_equal(np.isnan(warp), np.isnan(ref))
    assert np.array_equal(warp[finite], ref[finite])  # exact values while finite (infinities: same sign)
    assert np.array_equal(warp[np.isinf(ref)], ref[np.isinf(ref)])


@pytest.mark.parametrize("n", [64, 512])
def test_hierarchical_class_defaults_diverge_like_the_oracle(lsf, n):
    """100 iterations with the class defaults: the coarsest level hits the limit at |warp| ~ 1e15, the others stop at once"""
    c, l = O.sphere_pair(n, d=2)
    o, ref = _oracle_run(c, l)  # every keyword at its class default
    assert o.per_level_iteration_counts[0] == 100 and set(o.per_level_iteration_counts[1:]) == {1}
    assert np.isfinite(ref).all() and float(np.abs(ref).max()) > 1e12  # diverged, still finite
    for ci in (1, 32):
        opt = lsf.HierarchicalOptimizer2d(check_interval=ci)
        warp = opt.optimize(c, l)
        assert _compare_levels(opt, o) is None
        _compare_warps(warp, ref)
    # BASELINE config 2 as SURVEY 8(d) names it: chunk 4 => 3 levels n/4, n/2, n; strength 0.2, 100 iterations, 0.001
    kw = dict(maximum_chunk_size=4, tikhonov_strength=0.2, gradient_kernel_enabled=False, rate=0.1,
              maximum_iteration_count=100, maximum_warp_update_threshold=0.001)
    o, ref = _oracle_run(c, l, **kw)
    opt = lsf.HierarchicalOptimizer2d(**kw)
    warp = opt.optimize(c, l)
    assert len(o.per_level_iteration_counts) == 3
    _compare_levels(opt, o)
    _compare_warps(warp, ref)


@pytest.mark.parametrize("n", [64, 256])
def test_hierarchical_nan_regime_does_not_fault(lsf, n):
    """400 iterations: float32 overflows on the coarsest level (inf - inf = NaN).  No fault, the same counts, the same
    maxima up to the first non-finite iteration, the same NaN mask in the final warp"""
    c, l = O.sphere_pair(n, d=2)
    kw = dict(maximum_iteration_count=400)
    o, ref = _oracle_run(c, l, **kw)
    assert not np.isfinite(ref).all()
    opt = lsf.HierarchicalOptimizer2d(check_interval=50, **kw)
    warp = opt.optimize(c, l)
    first_bad = _compare_levels(opt, o)
    assert first_bad is not None and first_bad[0] == 0 and first_bad[1] > 100
    _compare_warps(warp, ref)
    torch.cuda.synchronize()  # a fault of the wild gathers would surface here at the latest


def test_hierarchical_3d_defaults_diverge_like_the_oracle(lsf):
    """the same in 3-D (4 D s = 2.4 per iteration): 32^3, class defaults but chunk 4 and 120 iterations"""
    c, l = O.sphere_pair(32, d=3)
    kw = dict(maximum_chunk_size=4, maximum_iteration_count=120)
    o, ref = _oracle_run(c, l, **kw)
    assert not np.isfinite(ref).all()
    opt = lsf.HierarchicalOptimizer3d(check_interval=40, **kw)
    warp = opt.optimize(c, l)
    assert _compare_levels(opt, o) is not None
    _compare_warps(warp, ref)
    torch.cuda.synchronize()


def test_slavcheva_leaves_through_the_upper_threshold(lsf, ref_slavcheva, tmp_path):
    """KillingFusion defaults on the reference's orthographic pair move 4-7 voxels per iteration; an upper threshold
    between two consecutive maxima ends the loop at exactly that iteration (slavcheva_optimizer2d.py:360-362)"""
    S = ref_slavcheva
    live0, canon = S["ortho64.live"], S["ortho64.canonical"]
    n = live0.shape[0]
    common = dict(level_set_term_enabled=True, maximum_warp_length_lower_threshold=0.05, max_iterations=12,
                  min_iterations=1)
    probe = O.SlavchevaOracle(compute_method=O.DIRECT, smoothing_term_method=O.KILLING, **common)
    probe.optimize(live0.copy(), canon)
    m = np.float32(probe.log["max_warps"])
    assert len(m) == 12
    # the first iteration k >= 2 whose maximum exceeds every earlier one: a threshold between ends the loop right after it
    k = next(i for i in range(2, len(m)) if m[i] > m[:i].max())
    hi = float((m[:k].max() + m[k]) / 2)
    for ci in (1, 4, 32):
        opt = lsf.SlavchevaOptimizer2d(out_path=str(tmp_path), field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                       maximum_warp_length_upper_threshold=hi, check_interval=ci, **common)
        live = live0.copy()
        opt.optimize(live, canon)
        ref = O.SlavchevaOracle(compute_method=O.DIRECT, smoothing_term_method=O.KILLING,
                                maximum_warp_length_upper_threshold=hi, **common)
        live_ref = live0.copy()
        ref.optimize(live_ref, canon)
        assert len(ref.log["max_warps"]) == k + 1 < 12           # iteration k crossed the threshold and was the last
        assert np.array_equal(np.float32(opt.log.max_warps), np.float32(ref.log["max_warps"]))
        assert np.array_equal(live, live_ref) and np.array_equal(opt.warp_field, ref.warp_field)
        report = opt.get_convergence_report()
        assert report.iteration_count == k + 1 and not report.iteration_limit_reached
        # the report's statistics are those of the FINAL warp field (slavcheva_optimizer2d.py:393-397), which the re-warp
        # has zeroed wherever the live field snapped (field_warping.py:138-141): its longest vector need not be the
        # iteration's maximum that ended the loop
        lengths = np.linalg.norm(opt.warp_field, axis=-1)
        assert abs(report.warp_delta_statistics.length_max - float(lengths.max())) <= 1e-5
        assert report.warp_delta_statistics.is_largest_above_max_threshold == bool(lengths.max() > hi)
