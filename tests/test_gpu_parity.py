"""Parity of the HIP path (through the C ABI of liblsf_hip.so) against the CPU oracle and the committed golden
fixtures.  Everything here needs a real MI355X: run with  pytest -m gpu.

Tolerances (absolute, float32 fields in [-1, 1], warps of a few voxels):
  * EXACT  = 0.0   -- the kernels follow the oracle's operation order with -ffp-contract=off, so everything that is
                      not a sum reduction is expected to be bit-identical; this is what the tests demand wherever
                      the oracle itself is the comparison target.
  * ATOL   = 1e-5  -- the north-star bound, used against the REFERENCE's goldens (numpy's own dot/convolve may
                      round differently from the oracle in the last place).
  * energies (sum reductions, float64 atomics): relative 1e-9.
"""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu

EXACT = 0.0
ATOL = 1e-5


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


def maxdiff(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def rand_field(rng, shape, noise=0.05):
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    f = 0.06 * (grids[-2] - shape[-2] / 2) + 0.3 * np.sin(grids[-1] * 0.35) + noise * rng.standard_normal(shape)
    if len(shape) == 3:
        f = f + 0.2 * np.cos(grids[0] * 0.4)
    return np.clip(f, -1.0, 1.0).astype(np.float32)


# ------------------------------------------------------------------------------------------ leaf kernels
@pytest.mark.parametrize("shape", [(12, 12), (33, 70), (9, 20, 67)])
def test_warp_field_kernels(lsf, shape):
    from levelsetfusion_python_amd.nonrigid_opt import field_warping as fw
    rng = np.random.default_rng(1)
    d = len(shape)
    field = rand_field(rng, shape)
    warp = (1.9 * rng.standard_normal(shape + (d,))).astype(np.float32)
    warp[(0,) * d] = -3.3
    warp[tuple(s - 1 for s in shape)] = 4.2
    assert maxdiff(fw.warp_field(field, warp), O.warp_field(field, warp)) == EXACT
    assert maxdiff(fw.warp_field_replacement(field, warp, 0.0), O.warp_field_replacement(field, warp, 0.0)) == EXACT
    assert maxdiff(fw.warp_field_replacement(field, warp, -0.5), O.warp_field_replacement(field, warp, -0.5)) == EXACT


@pytest.mark.parametrize("shape", [(12, 12), (10, 18, 66)])
@pytest.mark.parametrize("flags", [(False, False, False), (True, False, False), (False, True, False),
                                   (False, False, True), (True, True, True)])
def test_warp_field_advanced_kernel(lsf, shape, flags):
    from levelsetfusion_python_amd.nonrigid_opt import field_warping as fw
    rng = np.random.default_rng(2)
    d = len(shape)
    live = rand_field(rng, shape)
    canon = rand_field(rng, shape)
    warp = (0.6 * rng.standard_normal(shape + (d,))).astype(np.float32)
    wa, wb = warp.copy(), warp.copy()
    ga, gb = (wa * 10).astype(np.float32), (wb * 10).astype(np.float32)
    a = fw.warp_field_advanced(canon, live.copy(), wa, ga, *flags)
    b = O.warp_field_advanced(canon, live.copy(), wb, gb, *flags)
    assert maxdiff(a, b) == EXACT and maxdiff(wa, wb) == EXACT and maxdiff(ga, gb) == EXACT
    if flags == (False, False, False):
        assert (np.abs(b) == 1).any() and (wb == 0).any()  # the snap branch is exercised


def test_reference_field_warping_known_answers(lsf, ref_literals):
    from levelsetfusion_python_amd.nonrigid_opt import field_warping as fw
    T = ref_literals
    assert maxdiff(fw.warp_field(T["hierarchical_data.field_A_16x16"], T["hierarchical_data.warp_field_A_16x16"]),
                   T["hierarchical_data.fA_resampled_with_wfA"]) <= ATOL
    assert maxdiff(fw.warp_field_replacement(T["hierarchical_data.field_B_16x16"],
                                             T["hierarchical_data.warp_field_B_16x16"], 0.0),
                   T["hierarchical_data.fB_resampled_with_wfB_replacement"]) <= ATOL
    flags = {"01": (False, False, False), "02": (True, False, True), "03": (False, False, False),
             "04": (False, False, False), "05": (False, False, False)}
    for case, fl in flags.items():
        p = "field_warping.test_warp_field_advanced%s." % case
        warp = np.stack((T[p + "u_vectors"], T[p + "v_vectors"]), axis=2)
        grad = (warp * 10).astype(np.float32)
        new_live = fw.warp_field_advanced(T[p + "canonical_field"], T[p + "warped_live_template"].copy(), warp, grad,
                                          *fl)
        assert np.allclose(new_live, T[p + "expected_new_warped_live_field"], atol=ATOL), case
        if p + "expected_u_vectors" in T.files:
            assert np.allclose(warp[..., 0], T[p + "expected_u_vectors"], atol=ATOL), case
            assert np.allclose(warp[..., 1], T[p + "expected_v_vectors"], atol=ATOL), case


@pytest.mark.parametrize("shape", [(16, 64), (8, 16, 32)])
def test_pyramid_pack_restrict_prolong(lsf, shape):
    from levelsetfusion_python_amd import device as dev
    from levelsetfusion_python_amd.engine import as_device_field
    from levelsetfusion_python_amd.nonrigid_opt.hierarchical import pyramid
    rng = np.random.default_rng(3)
    d = len(shape)
    f = rand_field(rng, shape, 0.2)
    packed = dev.pack_live_gradient(as_device_field(f)).cpu().numpy()
    grads = O.gradient(f)
    assert maxdiff(packed[..., 0], f) == EXACT
    for c in range(d):
        assert maxdiff(packed[..., 1 + c], grads[c]) == EXACT
    cls = pyramid.ScalarFieldPyramid2d if d == 2 else pyramid.ScalarFieldPyramid3d
    levels = cls(f, 4).levels
    ref = O.pyramid(f, 4)
    assert len(levels) == len(ref) == 3
    for a, b in zip(levels, ref):
        assert a.shape == b.shape and maxdiff(a, b) == EXACT
    r4 = dev.restrict_mean(as_device_field(packed), 4).cpu().numpy()
    for c in range(4):
        assert maxdiff(r4[..., c], O.restrict_mean(np.ascontiguousarray(packed[..., c]))) == EXACT
    w = rng.standard_normal(shape + (d,)).astype(np.float32)
    up = dev.interleave(dev.prolong_repeat(dev.deinterleave(as_device_field(w), d))).cpu().numpy()
    assert maxdiff(up, O.prolong_repeat(w)) == EXACT
    with pytest.raises(ValueError):
        cls(np.zeros(tuple(s + 1 for s in shape), np.float32), 4)
    with pytest.raises(ValueError):
        cls(f, 3)


def test_reference_pyramid_known_answer(lsf):
    from levelsetfusion_python_amd.nonrigid_opt.hierarchical.pyramid import ScalarFieldPyramid2d
    tile = np.array([[1, 2, 5, 6, -1, -2, -5, -6], [3, 4, 7, 8, -3, -4, -7, -8],
                     [-1, -2, -5, -6, 1, 2, 5, 6], [-3, -4, -7, -8, 3, 4, 7, 8],
                     [1, 2, 5, 6, 5, 5, 5, 5], [3, 4, 7, 8, 5, 5, 5, 5],
                     [-1, -2, -5, -6, 5, 5, 5, 5], [-3, -4, -7, -8, 5, 5, 5, 5]], dtype=np.float32)
    levels = ScalarFieldPyramid2d(np.tile(tile, (16, 16))).levels  # tests/test_field_pyramid.py:24-73
    assert [l.shape for l in levels] == [(16, 16), (32, 32), (64, 64), (128, 128)]
    assert levels[2][0, 0] == tile[0:2, 0:2].mean() and levels[2][1, 1] == tile[2:4, 2:4].mean()
    assert levels[1][1, 1] == 5.0 and levels[0][0, 0] == 5.0 / 4


def test_convolution_kernels(lsf, ref_leaf, ref_literals):
    from levelsetfusion_python_amd.math_utils import convolution as mc
    L, T = ref_leaf, ref_literals
    vf2, vf3 = L["conv.vf2"], L["conv.vf3"]
    k7 = L["sobolev.hardcoded7"]
    for vf, kern in ((vf2, k7), (vf3, k7), (vf2, L["sobolev.k7"]), (vf2, np.array([0.5, 0.2, -0.1, 0.05, 0.3])),
                     (vf3, np.array([0.5, 0.2, -0.1])),
                     # even-length kernels: np.convolve's 'same' starts at (n - 1) // 2 (tests/test_oracle_golden.py)
                     (vf2, np.array([0.4, -0.2, 0.7, 0.1])), (vf3, np.array([1.0, 2.0]))):
        assert maxdiff(mc.convolve_with_kernel(vf.copy(), kern), O.convolve_with_kernel(vf.copy(), kern)) == EXACT
    assert maxdiff(mc.convolve_with_kernel_preserve_zeros(vf2.copy(), L["sobolev.k3"]),
                   O.convolve_with_kernel_preserve_zeros(vf2.copy(), L["sobolev.k3"])) == EXACT
    # against the reference's own outputs / known answers
    assert maxdiff(mc.convolve_with_kernel(vf2.copy(), k7), L["conv.vf2_k7f64"]) <= ATOL
    assert maxdiff(mc.convolve_with_kernel(vf3.copy(), k7), L["conv.vf3_k7f64"]) <= ATOL
    assert maxdiff(mc.convolve_with_kernel_preserve_zeros(vf2.copy(), k7), L["conv.vf2_pz_k7"]) <= ATOL
    v = np.arange(1.0, 241.0).reshape(5, 4, 4, 3).astype(np.float32)  # tests/test_convolution.py:215-219
    mc.convolve_with_kernel(v, np.array([3.0, 2.0, 1.0]))
    assert np.allclose(v, T["convolution_data.convolved_3d_vector_field"])
    v = np.arange(1.0, 241.0).reshape(5, 4, 4, 3).astype(np.float32)
    mc.convolve_with_kernel_x(v, np.array([3.0, 2.0, 1.0]))
    assert np.allclose(v, T["convolution_data.convolved_x_3d_vector_field"])
    p = "convolution.test_convolve_with_kernel_preserve_zeros02."
    v = T[p + "vector_field"].copy()
    mc.convolve_with_kernel_preserve_zeros(v, np.flip(T[p + "kernel"]))
    assert np.allclose(v, T[p + "expected_output"], rtol=0.0, atol=1e-6)
    with pytest.raises(ValueError):  # narrower than the kernel: the reference fails too
        mc.convolve_with_kernel(np.zeros((4, 16, 2), np.float32), k7)
    with pytest.raises(ValueError):
        mc.convolve_with_kernel(np.zeros((4, 16, 3), np.float32), k7)


def test_sobolev_kernel_generation(lsf, ref_leaf):
    for s, lam, k in ((3, 0.1, "k3"), (7, 0.1, "k7"), (9, 0.15, "k9")):
        assert maxdiff(lsf.generate_1d_sobolev_kernel(s, lam), ref_leaf["sobolev." + k]) <= 1e-7


# ------------------------------------------------------------------------------- hierarchical optimizer
def test_hierarchical_reference_golden_16x16(lsf, ref_literals):
    """tests/test_hierarchical_optimizer2d.py:39-54 of the reference"""
    T = ref_literals
    from levelsetfusion_python_amd.nonrigid_opt import field_warping as fw
    canon, live = T["hierarchical_data.canonical_field"], T["hierarchical_data.live_field"]
    canon0, live0 = canon.copy(), live.copy()
    opt = lsf.HierarchicalOptimizer2d(rate=0.2, data_term_amplifier=1.0, maximum_warp_update_threshold=0.001,
                                      maximum_iteration_count=100, tikhonov_term_enabled=False, kernel=None)
    warp = opt.optimize(canon, live)
    assert warp.shape == (16, 16, 2) and warp.dtype == np.float32
    assert np.array_equal(canon, canon0) and np.array_equal(live, live0)  # inputs untouched
    assert opt.get_per_level_iteration_counts() == [1, 100, 100, 100]
    assert maxdiff(warp, T["hierarchical_data.warp_field"]) <= ATOL
    assert maxdiff(fw.warp_field(live, warp), T["hierarchical_data.final_live_field"]) <= ATOL
    o = O.HierarchicalOracle(rate=0.2, maximum_warp_update_threshold=0.001, maximum_iteration_count=100,
                             tikhonov_term_enabled=False, kernel=None)
    assert maxdiff(warp, o.optimize(canon, live)) == EXACT


def test_hierarchical_runs_match_reference_and_oracle(lsf, ref_hierarchical, ref_literals):
    H, T = ref_hierarchical, ref_literals
    cases = {"g16": (T["hierarchical_data.canonical_field"], T["hierarchical_data.live_field"]),
             "c64": (H["c64.canonical"], H["c64.live"])}
    n = 0
    for key in H.files:
        if not key.endswith(".final_warp") or "threshold" in key:
            continue
        case, tik, ker, chunk, _ = key.split(".")
        kw = dict(tikhonov_term_enabled=tik == "tik1", gradient_kernel_enabled=ker == "ker1",
                  maximum_chunk_size=int(chunk[5:]), rate=0.2, maximum_iteration_count=4,
                  maximum_warp_update_threshold=0.0, tikhonov_strength=0.2,
                  kernel=H["kernel7"] if ker == "ker1" else None)
        warp = lsf.HierarchicalOptimizer2d(**kw).optimize(*cases[case])
        assert maxdiff(warp, H[key]) <= ATOL, key
        assert maxdiff(warp, O.HierarchicalOracle(**kw).optimize(*cases[case])) == EXACT, key
        n += 1
    assert n >= 10
    # threshold-terminated: the device-side gate must reproduce the reference's iteration counts for every
    # check interval (1 = sync every iteration ... 16 = sync once per 16)
    for ci in (1, 3, 16):
        opt = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=False,
                                          maximum_chunk_size=8, rate=0.1, maximum_iteration_count=40,
                                          maximum_warp_update_threshold=0.01, check_interval=ci)
        warp = opt.optimize(*cases["c64"])
        assert opt.get_per_level_iteration_counts() == list(H["c64.threshold_run.iteration_counts"])
        assert maxdiff(warp, H["c64.threshold_run.final_warp"]) <= ATOL


@pytest.mark.parametrize("tik,ker", [(False, False), (True, False), (False, True), (True, True)])
def test_hierarchical_threshold_gate_counts(lsf, tik, ker):
    """convergence gating with all kernel combinations: iteration counts and results equal the oracle's"""
    c, l = O.sphere_pair(64, d=2)
    # tikhonov_strength 0.05: the reference's Tikhonov recurrence g <- d - s*Laplace(g_prev) amplifies the highest
    # frequency by 4*D*s per iteration, i.e. it diverges for s >= 1/8 in 2-D (the default 0.2 does, after a few
    # dozen iterations); parity in a diverging run is meaningless, so the long runs use a stable strength
    kw = dict(tikhonov_term_enabled=tik, gradient_kernel_enabled=ker, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=60, maximum_warp_update_threshold=0.0295 if ker else 0.0245,
              tikhonov_strength=0.05, kernel=O.generate_1d_sobolev_kernel(7, 0.1) if ker else None)
    o = O.HierarchicalOracle(**kw)
    ref = o.optimize(c, l)
    assert 1 < o.per_level_iteration_counts[0] < 60  # the threshold, not the limit, ends level 0
    for ci in (1, 7):
        opt = lsf.HierarchicalOptimizer2d(check_interval=ci, **kw)
        warp = opt.optimize(c, l)
        assert opt.get_per_level_iteration_counts() == o.per_level_iteration_counts
        assert maxdiff(warp, ref) == EXACT
        for a, b in zip(opt.get_per_level_maximum_updates(), o.per_level_max_updates):
            assert np.array_equal(np.float32(a), np.float32(b))


@pytest.mark.parametrize("tik,ker", [(False, False), (True, False), (True, True)])
def test_hierarchical_3d_matches_oracle(lsf, tik, ker):
    c, l = O.sphere_pair(32, d=3)
    kw = dict(tikhonov_term_enabled=tik, gradient_kernel_enabled=ker, maximum_chunk_size=4, rate=0.2,
              maximum_iteration_count=3, maximum_warp_update_threshold=0.0, tikhonov_strength=0.2,
              kernel=O.generate_1d_sobolev_kernel(7, 0.1) if ker else None)
    warp = lsf.HierarchicalOptimizer3d(**kw).optimize(c, l)
    assert warp.shape == (32, 32, 32, 3)
    assert maxdiff(warp, O.HierarchicalOracle(**kw).optimize(c, l)) == EXACT


def test_hierarchical_errors(lsf):
    opt = lsf.HierarchicalOptimizer2d()
    with pytest.raises(ValueError):
        opt.optimize(np.zeros((12, 16), np.float32), np.zeros((12, 16), np.float32))
    with pytest.raises(ValueError):
        opt.optimize(np.zeros((8, 8), np.float32), np.zeros((8, 8), np.float32))  # chunk 8 too large
    with pytest.raises(ValueError):
        opt.optimize(np.zeros((16, 16), np.float32), np.zeros((32, 32), np.float32))
    # enable-flag folding (hierarchical_optimizer2d.py:96-107)
    o = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, tikhonov_strength=0.0, gradient_kernel_enabled=True,
                                    kernel=None)
    assert not o.tikhonov_term_enabled and not o.gradient_kernel_enabled


# ---------------------------------------------------------------------------------- Slavcheva optimizer
def _slavcheva_kwargs(lsf, name, S):
    CM, SM, DM = lsf.ComputeMethod, lsf.SmoothingTermMethod, lsf.DataTermMethod
    return {
        "sobolev_vec": (dict(compute_method=CM.VECTORIZED, sobolev_smoothing_enabled=True, sobolev_kernel=S["kernel7"]),
                        dict(compute_method=O.VECTORIZED, sobolev_smoothing_enabled=True, sobolev_kernel=S["kernel7"])),
        "sobolev_direct": (dict(compute_method=CM.DIRECT, sobolev_smoothing_enabled=True, sobolev_kernel=S["kernel3"]),
                           dict(compute_method=O.DIRECT, sobolev_smoothing_enabled=True, sobolev_kernel=S["kernel3"])),
        "killing": (dict(compute_method=CM.DIRECT, level_set_term_enabled=True, smoothing_term_method=SM.KILLING),
                    dict(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING)),
        "tikhonov_direct": (dict(compute_method=CM.DIRECT), dict(compute_method=O.DIRECT)),
        "fdm_direct": (dict(compute_method=CM.DIRECT, data_term_method=DM.THRESHOLDED_FDM),
                       dict(compute_method=O.DIRECT, data_term_method=O.THRESHOLDED_FDM)),
    }[name]


@pytest.mark.parametrize("size_tag,n_it", [("ortho32", 4), ("ortho64", 3)])
@pytest.mark.parametrize("name", ["sobolev_vec", "sobolev_direct", "killing", "tikhonov_direct", "fdm_direct"])
def test_slavcheva_2d_matches_reference_and_oracle(lsf, ref_slavcheva, tmp_path, size_tag, n_it, name):
    S = ref_slavcheva
    kw_gpu, kw_cpu = _slavcheva_kwargs(lsf, name, S)
    live0, canon = S[size_tag + ".live"], S[size_tag + ".canonical"]
    common = dict(maximum_warp_length_lower_threshold=0.0, max_iterations=n_it, min_iterations=n_it)
    opt = lsf.SlavchevaOptimizer2d(out_path=str(tmp_path), field_size=live0.shape[0], **common, **kw_gpu)
    live = live0.copy()
    assert opt.optimize(live, canon) is live  # warped in place and returned
    tag = size_tag + "." + name
    # against the reference's own run
    assert maxdiff(live, S[tag + ".final_live"]) <= ATOL
    assert maxdiff(opt.warp_field, S[tag + ".final_warp"]) <= ATOL
    assert maxdiff(opt.gradient_field, S[tag + ".final_gradient"]) <= ATOL
    assert maxdiff(opt.log.max_warps, S[tag + ".max_warps"]) <= ATOL
    for mine, theirs in ((opt.log.data_energies, S[tag + ".data_energies"]),
                         (opt.log.smoothing_energies, S[tag + ".smoothing_energies"]),
                         (opt.log.level_set_energies, S[tag + ".level_set_energies"])):
        assert np.allclose(mine, theirs, rtol=1e-5, atol=1e-7)
    # against the oracle: bit for bit
    o = O.SlavchevaOracle(**common, **kw_cpu)
    live_ref = live0.copy()
    o.optimize(live_ref, canon)
    assert maxdiff(live, live_ref) == EXACT
    assert maxdiff(opt.warp_field, o.warp_field) == EXACT
    assert maxdiff(opt.gradient_field, o.gradient_field) == EXACT
    assert np.array_equal(np.float32(opt.log.max_warps), np.float32(o.log["max_warps"]))
    assert opt.log.max_warp_locations == [tuple(int(i) for i in at[::-1]) for at in o.log["max_warp_locations"]]
    for mine, theirs in ((opt.log.data_energies, o.log["data_energies"]),
                         (opt.log.smoothing_energies, o.log["smoothing_energies"]),
                         (opt.log.level_set_energies, o.log["level_set_energies"])):
        assert np.allclose(mine, theirs, rtol=1e-9, atol=1e-12)


def test_slavcheva_reference_golden_4x4_and_report(lsf, ref_literals, ref_slavcheva, tmp_path):
    """tests/test_slavcheva_optimizer.py:64-149 of the reference (live-field goldens + convergence report)"""
    from levelsetfusion_python_amd.convergence_report import (ConvergenceReport, TsdfDifferenceStatistics,
                                                              WarpDeltaStatistics)
    T = ref_literals
    live0 = T["slavcheva.test_nonrigid_optimization01.live_field_template"]
    canon = T["slavcheva.test_nonrigid_optimization01.canonical_field"]
    reports = []
    for method in (lsf.ComputeMethod.VECTORIZED, lsf.ComputeMethod.DIRECT):
        for n_it, key in ((1, "slavcheva.test_nonrigid_optimization01.expected_live_field_out"),
                          (2, "slavcheva.test_nonrigid_optimization02.expected_live_field_out")):
            opt = lsf.SlavchevaOptimizer2d(out_path=str(tmp_path), field_size=4, compute_method=method,
                                           sobolev_smoothing_enabled=True, maximum_warp_length_lower_threshold=0.05,
                                           max_iterations=n_it, sobolev_kernel=ref_slavcheva["kernel3"])
            live = live0.copy()
            opt.optimize(live, canon)
            assert np.allclose(live, T[key], atol=ATOL)
        reports.append(opt.get_convergence_report())
    assert reports[0] == reports[1]
    ws, ts = T["slavcheva.report.warp_stats"], T["slavcheva.report.tsdf_stats"]
    expected = ConvergenceReport(2, True,
                                 WarpDeltaStatistics(ws[0], ws[1], ws[2], ws[3], ws[4], (int(ws[5]), int(ws[6])),
                                                     False, False),
                                 TsdfDifferenceStatistics(ts[0], ts[1], ts[2], ts[3], (int(ts[4]), int(ts[5]))))
    assert reports[1] == expected


@pytest.mark.parametrize("ci", [1, 4, 32])
def test_slavcheva_threshold_termination(lsf, ref_slavcheva, tmp_path, ci):
    """loop condition of slavcheva_optimizer2d.py:360-362 on the device-side gate, any check interval"""
    S = ref_slavcheva
    live0, canon = S["ortho32.live"], S["ortho32.canonical"]
    kw = dict(maximum_warp_length_lower_threshold=0.1, max_iterations=25, min_iterations=2)
    o = O.SlavchevaOracle(compute_method=O.DIRECT, **kw)
    live_ref = live0.copy()
    o.optimize(live_ref, canon)
    assert 2 < o.iteration_count < 25  # the threshold, not a limit, ends this run
    opt = lsf.SlavchevaOptimizer2d(out_path=str(tmp_path), field_size=32, compute_method=lsf.ComputeMethod.DIRECT,
                                   check_interval=ci, **kw)
    live = live0.copy()
    opt.optimize(live, canon)
    assert len(opt.log.max_warps) == o.iteration_count
    assert maxdiff(live, live_ref) == EXACT and maxdiff(opt.warp_field, o.warp_field) == EXACT
    assert opt.get_convergence_report().iteration_count == o.iteration_count


def test_slavcheva_errors(lsf, tmp_path):
    opt = lsf.SlavchevaOptimizer2d(out_path=str(tmp_path / "made"), field_size=8)
    assert (tmp_path / "made").exists()
    with pytest.raises(ValueError):
        opt.optimize(np.zeros((8, 8), np.float32), np.zeros((8, 4), np.float32))
    with pytest.raises(ValueError):
        opt.optimize(np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))  # field_size mismatch


@pytest.mark.parametrize("name", ["killing", "sobolev_direct", "sobolev_vec", "fdm_direct"])
def test_slavcheva_3d_matches_oracle(lsf, ref_slavcheva, name):
    kw_gpu, kw_cpu = _slavcheva_kwargs(lsf, name, ref_slavcheva)
    n = 32
    canon, live0 = O.sphere_pair(n, d=3)
    common = dict(maximum_warp_length_lower_threshold=0.0, max_iterations=3, min_iterations=3)
    opt = lsf.SlavchevaOptimizer3d(field_size=n, **common, **kw_gpu)
    live = live0.copy()
    opt.optimize(live, canon)
    o = O.SlavchevaOracle(**common, **kw_cpu)
    live_ref = live0.copy()
    o.optimize(live_ref, canon)
    assert maxdiff(live, live_ref) == EXACT
    assert maxdiff(opt.warp_field, o.warp_field) == EXACT
    assert maxdiff(opt.gradient_field, o.gradient_field) == EXACT
    for mine, theirs in ((opt.log.data_energies, o.log["data_energies"]),
                         (opt.log.smoothing_energies, o.log["smoothing_energies"]),
                         (opt.log.level_set_energies, o.log["level_set_energies"])):
        assert np.allclose(mine, theirs, rtol=1e-9, atol=1e-12)


def test_slavcheva_3d_random_fields_match_oracle(lsf):
    """non-smooth inputs with truncated plateaus, OOB gathers and snapping"""
    rng = np.random.default_rng(11)
    shape = (24, 24, 24)
    live0 = rand_field(rng, shape, 0.15)
    canon = rand_field(rng, shape, 0.15)
    live0[:, :5] = 1.0
    canon[:, :4] = 1.0
    live0[3:6, 10:14, 10:14] = -1.0
    common = dict(maximum_warp_length_lower_threshold=0.0, max_iterations=3, min_iterations=3,
                  gradient_descent_rate=0.1)
    opt = lsf.SlavchevaOptimizer3d(field_size=24, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING, **common)
    live = live0.copy()
    opt.optimize(live, canon)
    o = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
                          **common)
    live_ref = live0.copy()
    o.optimize(live_ref, canon)
    assert maxdiff(live, live_ref) == EXACT and maxdiff(opt.warp_field, o.warp_field) == EXACT


@pytest.mark.parametrize("shape,zr", [((12, 20, 200), None), ((5, 3, 7), None), ((9, 33, 65), (2, 7)),
                                      ((1, 40, 70), None), ((6, 8, 8), (3, 3))])
def test_band_list_matches_numpy(lsf, shape, zr):
    """lsf_band_count / lsf_band_list_fill: ascending indices of the voxels with |live| != 1 or |canonical| != 1
    (tsdf_set_routines.py:19-52) inside the z-range, ragged sizes, 2-D, empty range, all / none in band"""
    from levelsetfusion_python_amd import _lib, device as dev
    rng = np.random.default_rng(5)
    for fill in ("mixed", "none", "all"):
        live = rng.uniform(-1, 1, shape).astype(np.float32)
        canon = rng.uniform(-1, 1, shape).astype(np.float32)
        if fill != "all":
            live[rng.uniform(size=shape) < (0.7 if fill == "mixed" else 2.0)] = 1.0
            canon[rng.uniform(size=shape) < (0.7 if fill == "mixed" else 2.0)] = -1.0
        if shape[0] == 1:
            grid = dev.make_grid(shape[1:])
            lo, hi = 0, 1
        else:
            lo, hi = zr if zr else (0, shape[0])
            grid = dev.make_grid(shape, lo, hi, 0)
        t_shape = shape if shape[0] > 1 else shape[1:]
        tl, tc = torch.from_numpy(live.reshape(t_shape)).cuda(), torch.from_numpy(canon.reshape(t_shape)).cuda()
        mask = ~((np.abs(live) == 1) & (np.abs(canon) == 1))
        mask[:lo] = False
        mask[hi:] = False
        inner = np.zeros(shape, bool)
        inner[(slice(1, -1),) * 3 if shape[0] > 1 else (slice(None), slice(1, -1), slice(1, -1))] = True
        for subset, want in ((_lib.BAND_ALL, mask), (_lib.BAND_INTERIOR, mask & inner), (_lib.BAND_BOUNDARY, mask & ~inner)):
            band = dev.band_list(tl, tc, grid, subset)
            expect = np.flatnonzero(want.ravel())
            assert band.count == len(expect) and band.subset == subset
            assert np.array_equal(band.indices.cpu().numpy()[:band.count], expect)
        lists = dev.band_lists(tl, tc, grid)
        assert 1 <= len(lists) <= 2 and sum(b.count for b in lists) == int(mask.sum())
        assert all(b.count > 0 for b in lists) or (len(lists) == 1 and lists[0].count == 0)


def test_slavcheva_band_list_is_invisible(lsf):
    """lsf_slavcheva_state_iteration with a band list visits only the listed voxels; with both ping-pong states
    initialised as the header demands, states and records must equal those of the dense walk -- 200-voxel lines, voxels
    that leave the band by snapping, and the oracle as third opinion"""
    from levelsetfusion_python_amd import _lib, device as dev
    n_it = 9
    shape = (12, 20, 200)
    canon, live0 = O.sphere_pair(32, 3)
    canon = np.ascontiguousarray(np.tile(canon[10:22, 6:26, :], (1, 1, 7))[:, :, :200])
    live0 = np.ascontiguousarray(np.tile(live0[10:22, 6:26, :], (1, 1, 7))[:, :, :200])
    live0[:, :, 130:] = 1.0
    canon[:, :, 130:] = -1.0
    eng = lsf.SlavchevaOptimizer3d(field_size=32, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                   gradient_descent_rate=0.5).engine
    grid = dev.make_grid(shape)
    outs = []
    for mode in ("dense", "one list", "interior + boundary"):
        c = torch.from_numpy(canon).cuda()
        l0 = torch.from_numpy(live0).cuda()
        states = dev.state_pack(l0, None, grid, copies=2)
        rec = dev.new_records(n_it, "cuda")
        bands = [None] if mode == "dense" else dev.band_lists(l0, c, grid, split=mode != "one list")
        if mode == "one list":
            assert len(bands) == 1 and 0 < bands[0].count < live0.size * 0.7
        if mode == "interior + boundary":
            assert [b.subset for b in bands] == [_lib.BAND_INTERIOR, _lib.BAND_BOUNDARY]
        for i in range(n_it):
            for band in bands:
                dev.slavcheva_state_iteration(states[i % 2], c, states[(i + 1) % 2], grid, eng.params, None, rec, i,
                                              band)
        outs.append([t.cpu().numpy() for t in states + [rec]])
    for other in outs[1:]:
        for a, b in zip(outs[0][:2], other[:2]):
            assert np.array_equal(a, b)
        ra, rb = dev.decode_records(outs[0][2]), dev.decode_records(other[2])
        assert np.array_equal(ra["max_value"], rb["max_value"]) and np.array_equal(ra["argmax"], rb["argmax"])
        for k in ("data_energy", "smoothing_energy", "level_set_energy"):
            # the energy sums are float64 atomics whose order varies from launch to launch
            assert np.allclose(ra[k], rb[k], rtol=1e-9, atol=1e-12) and np.all(ra[k][1:] > 0)
    o = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
                          gradient_descent_rate=0.5, max_iterations=n_it, min_iterations=n_it,
                          maximum_warp_length_lower_threshold=0.0)
    live_ref = live0.copy()
    o.optimize(live_ref, canon)
    assert maxdiff(outs[2][n_it % 2][..., 0], live_ref) == EXACT
    assert maxdiff(outs[2][n_it % 2][..., 1:], o.warp_field) == EXACT


def test_slavcheva_band_list_all_zero_update_reports_first_voxel(lsf):
    """nothing in the band: every update has length 0 and np.argmax reports voxel 0 (slavcheva_optimizer2d.py:222-224);
    the list is empty and the kernel must still say so"""
    from levelsetfusion_python_amd import _lib, device as dev
    shape = (4, 8, 70)
    ones = torch.ones(shape, device="cuda")
    eng = lsf.SlavchevaOptimizer3d(field_size=8).engine
    grid = dev.make_grid(shape, 1, 3, 5)
    band = dev.band_list(ones, ones, grid)
    assert band.count == 0
    rec = dev.new_records(1, "cuda")
    states = dev.state_pack(ones, None, grid, copies=2)
    dev.slavcheva_state_iteration(states[0], ones, states[1], grid, eng.params, None, rec, 0, band)
    dec = dev.decode_records(rec.cpu().numpy())
    assert dec["max_value"][0] == 0.0 and dec["argmax"][0] == (1 + 5) * 8 * 70


def test_hierarchical_energy_printouts(lsf, capsys):
    """VerbosityParameters(print_iteration_data_energy, print_iteration_tikhonov_energy): the sums behind the reference's
    per-iteration printouts (hierarchical_optimizer2d.py:204-210, 233-238) are accumulated by the iteration kernel; 2-D
    (graph-replayed levels) and 3-D against the oracle's float64 sums, and the printed line format"""
    for d, n in ((2, 64), (3, 32)):
        canon, live = O.sphere_pair(n, d)
        kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8, rate=0.1,
                  maximum_iteration_count=6, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
        cls = lsf.HierarchicalOptimizer2d if d == 2 else lsf.HierarchicalOptimizer3d
        vp = cls.VerbosityParameters(print_max_warp_update=True, print_iteration_data_energy=True,
                                     print_iteration_tikhonov_energy=True)
        opt = cls(verbosity_parameters=vp, **kw)
        warp = opt.optimize(canon, live)
        o = O.HierarchicalOracle(**kw)
        assert maxdiff(warp, o.optimize(canon, live)) == EXACT
        results = opt.engine.level_results
        assert len(results) == len(o.per_level_data_energy_sums)
        for r, want_data, want_tik in zip(results, o.per_level_data_energy_sums, o.per_level_tikhonov_energy_sums):
            assert np.allclose(r.data_energies, want_data, rtol=1e-9, atol=0.0)
            assert np.allclose(r.tikhonov_energies, want_tik, rtol=1e-6, atol=1e-30)
            assert r.tikhonov_energies[0] == 0.0 and r.tikhonov_energies[-1] > 0.0  # zero previous gradient at level start
        printed = capsys.readouterr().out
        last = results[-1]
        want = " norm. tikhonov energy: %f" % (1000000.0 * 0.5 * last.tikhonov_energies[-1] / last.voxel_count)
        assert want in printed and " norm. data energy: " in printed and " max upd. l.: " in printed


def test_state_pack_unpack_finalize(lsf):
    """lsf_state_pack / lsf_state_unpack round trip (2-D and 3-D, ragged extents, z-ranges) and lsf_state_finalize:
    fields equal the unpacked ones, statistics equal those of the two stand-alone statistics kernels (a20)"""
    from levelsetfusion_python_amd import device as dev
    gen = torch.Generator("cuda").manual_seed(11)
    for shape in ((7, 13, 70), (33, 130), (4, 8, 64)):
        dims = len(shape)
        live = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
        canon = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
        live[torch.rand(shape, device="cuda", generator=gen) < 0.4] = 1.0
        canon[torch.rand(shape, device="cuda", generator=gen) < 0.6] = -1.0
        warp = torch.randn((dims,) + shape, device="cuda", generator=gen)
        grid = dev.make_grid(shape)
        a, b = dev.state_pack(live, warp, grid, copies=2)
        assert torch.equal(a, b) and torch.equal(a[..., 0], live)
        for c in range(dims):
            assert torch.equal(a[..., 1 + c], warp[c])
        if dims == 2:
            assert float(a[..., 3].abs().max()) == 0.0
        z0 = dev.state_pack(live, None, grid, copies=1)[0]
        assert float(z0[..., 1:].abs().max()) == 0.0
        # lsf_state_prepare = lsf_state_pack (two copies, zero warp) + the INTERIOR / BOUNDARY lists of lsf_band_count
        st, lists, unlisted = dev.state_prepare(live, canon, grid)
        assert torch.equal(st[0], z0) and torch.equal(st[1], z0)
        outside = (live.abs() == 1) & (canon.abs() == 1) & (live != canon)
        assert unlisted[0] == int(outside.sum())
        assert unlisted[1] == (int(outside.flatten().nonzero()[0]) if unlisted[0] else -1)
        want_lists = dev.band_lists(live, canon, grid)
        assert [(bl.subset, bl.count) for bl in lists] == [(bl.subset, bl.count) for bl in want_lists]
        for got_l, want_l in zip(lists, want_lists):
            assert torch.equal(got_l.indices[:got_l.count], want_l.indices[:want_l.count])
        l2, p2, i2 = torch.empty_like(live), torch.empty_like(warp), torch.empty(shape + (dims,), device="cuda")
        dev.state_unpack(a, grid, l2, p2, i2)
        assert torch.equal(l2, live) and torch.equal(p2, warp) and torch.equal(i2, dev.interleave(warp))
        # finalize on a z-range (3-D) / the whole field
        g = dev.make_grid(shape, 1, shape[0] - 2, 3) if dims == 3 else grid
        l3 = torch.zeros_like(live)
        i3 = torch.zeros(shape + (dims,), device="cuda")
        raw = dev.state_finalize(a, canon, g, l3, None, i3, 0.25, True).cpu().numpy()
        zs = slice(g.z_begin, g.z_end) if dims == 3 else slice(None)
        assert torch.equal(l3[zs], live[zs]) and torch.equal(i3[zs], dev.interleave(warp)[zs])
        if dims == 3:
            assert float(l3[:1].abs().max()) == 0.0 and float(i3[-2:].abs().max()) == 0.0
        # the same pass from planar final fields (the SobolevFusion path's layout): identical in every bit
        l4, i4 = torch.zeros_like(live), torch.zeros(shape + (dims,), device="cuda")
        raw_planar = dev.planar_finalize(live, warp, canon, g, l4, i4, 0.25, True).cpu().numpy()
        assert torch.equal(l4, l3) and torch.equal(i4, i3) and np.array_equal(raw_planar, raw)
        want_w = dev.warp_statistics(warp, canon, live, 0.25, g).cpu().numpy()
        want_d = dev.tsdf_difference_statistics(canon, live, g).cpu().numpy()
        for got, want in ((raw[:8], want_w), (raw[8:], want_d)):
            assert np.array_equal(got[[0, 1, 2, 5]], want[[0, 1, 2, 5]])      # counts, max / min, arg-max: exact
            assert np.allclose(got[3:5], want[3:5], rtol=1e-12, atol=0.0)     # float64 sums in a different order


def test_fused_xy_filter_equals_two_passes(lsf):
    """lsf_convolve_xy (x and y pass of a 3-D filter in one launch, slice by slice) == lsf_convolve_axis along x, then along
    y, bit for bit: ragged extents, every tap count, float32-valued and arbitrary taps, a z-range (slices outside it are not
    written), a closed gate"""
    from levelsetfusion_python_amd import _lib, device as dev
    gen = torch.Generator("cuda").manual_seed(5)
    for shape in ((37, 35, 72), (64, 64, 64), (9, 10, 12), (5, 130, 132), (16, 16, 16)):
        grids = [dev.make_grid(shape)] + ([dev.make_grid(shape, 4, shape[0] - 2, 9)] if shape[0] > 8 else [])
        for grid in grids:
            for n_taps in (3, 5, 7, 9):
                taps = lsf.generate_1d_sobolev_kernel(n_taps, 0.1) if n_taps in (3, 7) else \
                    np.linspace(-0.2, 1.0, n_taps).astype(np.float64) / 3.0
                assert dev.convolve_xy_ok(grid, taps)
                for planes in (1, 3):
                    src = torch.randn((planes,) + shape, device="cuda", generator=gen)
                    a, want, got = torch.full_like(src, 5.0), torch.full_like(src, 7.0), torch.full_like(src, 7.0)
                    dev.convolve_axis(src, a, None, grid, 0, taps)
                    dev.convolve_axis(a, want, None, grid, 1, taps)
                    dev.convolve_xy(src, got, grid, taps)
                    assert torch.equal(got, want), (shape, n_taps, planes, float((got - want).abs().max()))
    shape = (16, 16, 64)
    grid = dev.make_grid(shape)
    rec = dev.new_records(2, "cuda")
    dev.set_record_max(rec, 0, int(np.float32(0.001).view(np.uint32)) << 32 | 5)
    gate = _lib.Gate(rec.data_ptr(), _lib.GATE_HIERARCHICAL, 0.01, 0.0)
    src, out = torch.randn((3,) + shape, device="cuda", generator=gen), torch.full((3,) + shape, 7.0, device="cuda")
    dev.convolve_xy(src, out, grid, lsf.generate_1d_sobolev_kernel(7, 0.1), gate)
    assert bool((out == 7.0).all())
    assert not dev.convolve_xy_ok(dev.make_grid((16, 16, 18)), np.ones(7))  # nx % 4
    assert not dev.convolve_xy_ok(dev.make_grid((16, 16)), np.ones(7))      # 2-D: the reference's order is y, then x


def test_last_filter_pass_that_moves_the_warp(lsf):
    """lsf_convolve_axis_update (the last pass of a hierarchical iteration's filter, which also moves the warp) == the plain
    pass + lsf_hier_update's warp half, bit for bit: along z in 3-D (what the engine uses) and along y in 2-D, ragged extents, every tap count, a
    z-range, float32-valued and arbitrary taps; a closed gate leaves gradient and warp alone"""
    from levelsetfusion_python_amd import _lib, device as dev
    gen = torch.Generator("cuda").manual_seed(11)
    for shape in ((37, 35, 72), (64, 64, 64), (9, 10, 12), (70, 130), (33, 47)):
        dims = len(shape)
        grids = [dev.make_grid(shape)] + ([dev.make_grid(shape, 5, shape[0] - 3, 11)] if dims == 3 and shape[0] > 12 else [])
        for grid in grids:
            for n_taps in (3, 5, 7, 9):
                taps = lsf.generate_1d_sobolev_kernel(n_taps, 0.1) if n_taps in (3, 7) else \
                    np.linspace(-0.2, 1.0, n_taps).astype(np.float64) / 3.0  # (not float32 values: the other arithmetic)
                assert dev.convolve_axis_update_ok(grid, taps) == (dims == 3)
                src = torch.randn((dims,) + shape, device="cuda", generator=gen)
                warp0 = torch.randn((dims,) + shape, device="cuda", generator=gen)
                want_g, want_w = torch.full_like(src, 7.0), warp0.clone()
                dev.convolve_axis(src, want_g, None, grid, dims - 1, taps)
                rec = dev.new_records(1, src.device)
                dev.hier_update(want_g, want_w, grid, 0.3, None, rec, 0)
                got_g, got_w = torch.full_like(src, 7.0), warp0.clone()
                dev.convolve_axis_update(src, got_g, got_w, 0.3, grid, dims - 1, taps)
                assert torch.equal(got_g, want_g) and torch.equal(got_w, want_w), (shape, n_taps)
                assert not torch.equal(got_w, warp0)
    # a closed gate (the previous iteration converged): nothing is written
    shape = (16, 16, 64)
    grid = dev.make_grid(shape)
    rec = dev.new_records(2, "cuda")
    dev.set_record_max(rec, 0, int(np.float32(0.001).view(np.uint32)) << 32 | 5)
    gate = _lib.Gate(rec.data_ptr(), _lib.GATE_HIERARCHICAL, 0.01, 0.0)
    src, warp = torch.randn((3,) + shape, device="cuda", generator=gen), torch.randn((3,) + shape, device="cuda", generator=gen)
    g_out, w0 = torch.full_like(src, 7.0), warp.clone()
    dev.convolve_axis_update(src, g_out, warp, 0.3, grid, 2, lsf.generate_1d_sobolev_kernel(7, 0.1), gate)
    assert torch.equal(warp, w0) and bool((g_out == 7.0).all())
    # what the entry point does not implement
    with pytest.raises(Exception):
        dev.convolve_axis_update(src, g_out, warp, 0.3, grid, 2, np.ones(4))  # four taps
    assert not dev.convolve_axis_update_ok(grid, np.ones(4))


def test_fused_xyz_filter_equals_three_passes(lsf):
    """lsf_convolve_xyz (x, y and z pass in one launch) == three lsf_convolve_axis passes, bit for bit: ragged
    extents (tiles, rows and z chunks that end early), every supported tap count, a closed gate leaves dst alone"""
    from levelsetfusion_python_amd import device as dev
    gen = torch.Generator("cuda").manual_seed(3)
    for shape in ((37, 35, 72), (64, 64, 64), (9, 10, 12), (70, 18, 132)):
        grid = dev.make_grid(shape)
        for planes in (1, 4):  # a scalar field / the four channels of a packed field
            one = torch.randn((planes,) + shape, device="cuda", generator=gen)
            k5 = np.linspace(-0.2, 1.0, 5).astype(np.float32)
            a1, b1, f1 = torch.empty_like(one), torch.empty_like(one), torch.empty_like(one)
            dev.convolve_axis(one, a1, None, grid, 0, k5)
            dev.convolve_axis(a1, b1, None, grid, 1, k5)
            dev.convolve_axis(b1, a1, None, grid, 2, k5)
            dev.convolve_xyz(one, f1, grid, k5)
            assert torch.equal(f1, a1), (shape, planes)
        src = torch.randn((3,) + shape, device="cuda", generator=gen)
        src[:, :, :, : shape[2] // 3] = 0.0  # exact zeros as in a gradient that vanishes outside a band
        for n_taps in (3, 5, 7, 9):
            taps = lsf.generate_1d_sobolev_kernel(n_taps, 0.1) if n_taps in (3, 7) else \
                np.linspace(-0.2, 1.0, n_taps).astype(np.float32)
            assert dev.convolve_xyz_ok(grid, taps)
            a, b = torch.empty_like(src), torch.empty_like(src)
            dev.convolve_axis(src, a, None, grid, 0, taps)
            dev.convolve_axis(a, b, None, grid, 1, taps)
            dev.convolve_axis(b, a, None, grid, 2, taps)
            fused = torch.full_like(src, 7.0)
            dev.convolve_xyz(src, fused, grid, taps)
            assert torch.equal(fused, a), (shape, n_taps, float((fused - a).abs().max()))
            # with the hierarchical update folded in: warp -= rate * filtered, and lsf_hier_update(warp = NULL) then only
            # leaves the maximum in the record
            warp0 = torch.randn((3,) + shape, device="cuda", generator=gen)
            w_ref, w_fused = warp0.clone(), warp0.clone()
            rec = dev.new_records(2, src.device)
            dev.hier_update(a, w_ref, grid, 0.3, None, rec, 0)
            fused.fill_(7.0)
            dev.convolve_xyz(src, fused, grid, taps, None, w_fused, 0.3)
            dev.hier_update(fused, None, grid, 0.3, None, rec, 1)
            assert torch.equal(fused, a) and torch.equal(w_fused, w_ref)
            dec = dev.decode_records(dev.records_to_host(rec))
            assert dec["max_value"][0] == dec["max_value"][1] > 0 and dec["argmax"][0] == dec["argmax"][1]
    # a z-range (what a z-slab filters: its owned slices, from raw input whose halo slices are valid): x and y passes on
    # every slice, the z pass on the range; slices outside the range are not written
    shape = (40, 18, 68)
    src = torch.randn((3,) + shape, device="cuda", generator=gen)
    taps = lsf.generate_1d_sobolev_kernel(7, 0.1)
    whole, part = dev.make_grid(shape), dev.make_grid(shape, 5, 33, 11)
    a, b = torch.empty_like(src), torch.empty_like(src)
    dev.convolve_axis(src, a, None, whole, 0, taps)
    dev.convolve_axis(a, b, None, whole, 1, taps)
    want = torch.full_like(src, 7.0)
    dev.convolve_axis(b, want, None, part, 2, taps)
    fused = torch.full_like(src, 7.0)
    dev.convolve_xyz(src, fused, part, taps)
    assert torch.equal(fused, want) and float(fused[:, :5].min()) == 7.0 and float(fused[:, 33:].max()) == 7.0
    assert not dev.convolve_xyz_ok(dev.make_grid((16, 16, 18)), np.ones(7))   # nx % 4
    assert not dev.convolve_xyz_ok(dev.make_grid((16, 16, 16)), np.ones(4))   # even tap count
    assert not dev.convolve_xyz_ok(dev.make_grid((4, 16, 16)), np.ones(7))    # shorter than the kernel
    with pytest.raises(ValueError):
        dev.convolve_xyz(src, fused, dev.make_grid((16, 16)), np.ones(7))


def test_state_prepare_reports_list_positions_of_chunk_boundaries(lsf):
    """StatePrepare(cut_chunks=...): the number of INTERIOR / BOUNDARY list entries in front of voxel 1024 * chunk comes
    back with the list sizes -- equal to a search of the filled lists (what a z-slab run otherwise does for its z cuts)"""
    from levelsetfusion_python_amd import _lib, device as dev
    gen = torch.Generator("cuda").manual_seed(17)
    shape = (24, 32, 64)  # 2048 voxels per slice: two chunks
    live = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
    canon = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
    far = torch.rand(shape, device="cuda", generator=gen) < 0.7
    live[far], canon[far] = 1.0, -1.0
    live[5:9] = 1.0
    canon[5:9] = 1.0  # a run of slices without any band voxel
    grid = dev.make_grid(shape)
    chunks = dev.n_voxels(grid) // 1024
    picks = torch.tensor([0, 1, 2, 7, 10, 11, 18, chunks - 1, chunks], dtype=torch.int64, device="cuda")
    prepared = dev.StatePrepare(live, canon, grid, picks)
    lists, _ = prepared.collect()
    assert len(lists) == 2 and prepared.cuts is not None
    for bl in lists:
        want = torch.searchsorted(bl.indices[:bl.count], (picks * 1024).to(torch.int32)).tolist()
        got = prepared.cuts[bl.subset]
        assert got[:-1] == want[:-1]                      # a chunk index of `chunks` is the end: cut_totals
        assert prepared.cut_totals[bl.subset] == bl.count == want[-1]
    plain = dev.StatePrepare(live, canon, grid)
    plain.collect()
    assert plain.cuts is None


@pytest.mark.gpu
def test_state_prepare_second_state_behind_the_sizes_or_in_the_pass(lsf, monkeypatch):
    """the second ping-pong state is written either by the counting pass itself (above StatePrepare.SPLIT_MAX_VOXELS) or by
    lsf_state_pack behind the copy of the list sizes: same states, same lists either way"""
    from levelsetfusion_python_amd import device as dev
    gen = torch.Generator("cuda").manual_seed(23)
    shape = (20, 24, 40)
    live = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
    canon = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
    far = torch.rand(shape, device="cuda", generator=gen) < 0.6
    live[far], canon[far] = -1.0, 1.0
    results = []
    for limit in (dev.StatePrepare.SPLIT_MAX_VOXELS, 0):
        monkeypatch.setattr(dev.StatePrepare, "SPLIT_MAX_VOXELS", limit)
        prepared = dev.StatePrepare(live, canon)
        lists, unlisted = prepared.collect()
        torch.cuda.synchronize()
        results.append((prepared.states, lists, unlisted))
    want = torch.zeros(shape + (4,), device="cuda")
    want[..., 0] = live
    for states, lists, unlisted in results:
        assert torch.equal(states[0], want) and torch.equal(states[1], want)
        assert unlisted == results[0][2]
        for a, b in zip(lists, results[0][1]):
            assert a.count == b.count and torch.equal(a.indices[:a.count], b.indices[:b.count])
    monkeypatch.setattr(dev.StatePrepare, "SPLIT_MAX_VOXELS", 1)
    big = dev.StatePrepare(live, canon)
    big.collect()
    torch.cuda.synchronize()
    assert torch.equal(big.states[1], want)


def test_state_finalize_listed_equals_dense_finalize(lsf):
    """lsf_state_finalize_listed (band voxels only + lsf_state_prepare's counts of the rest) against lsf_state_finalize
    over every voxel: fields bit-identical, counts / extrema / arg-max exact, float64 sums to rounding.  Cases: a band
    with both kinds of outside voxels, no outside voxel with live = -canonical, an empty band, identical fields."""
    from levelsetfusion_python_amd import device as dev
    gen = torch.Generator("cuda").manual_seed(5)

    def fields(shape, case):
        live = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
        canon = (torch.rand(shape, device="cuda", generator=gen) * 2 - 1).contiguous()
        if case == "identical":
            live[torch.rand(shape, device="cuda", generator=gen) < 0.5] = 1.0
            return live, live.clone()
        if case == "empty band":
            live = torch.where(live > 0, 1.0, -1.0).contiguous()
            canon = torch.where(canon > 0, 1.0, -1.0).contiguous()
            return live, canon
        far = torch.rand(shape, device="cuda", generator=gen) < 0.6
        sign = torch.where(torch.rand(shape, device="cuda", generator=gen) < 0.5, 1.0, -1.0)
        live[far], canon[far] = sign[far], sign[far]
        if case == "both":
            flip = far & (torch.rand(shape, device="cuda", generator=gen) < 0.3)
            flip.view(-1)[:3] = False  # the first opposite voxel is not the first voxel
            canon[flip] = -canon[flip]
        return live, canon

    for shape in ((9, 21, 67), (45, 131)):
        dims = len(shape)
        for case in ("both", "no opposite", "empty band", "identical"):
            live, canon = fields(shape, case)
            grid = dev.make_grid(shape)
            st, lists, unlisted = dev.state_prepare(live, canon, grid)
            state = st[0]
            flat = state.view(-1, 4)
            for bl in lists:  # move the listed voxels the way iterations would: new live values (some to +-1), a warp
                if bl.count == 0:
                    continue
                idx = bl.indices[:bl.count].long()
                vals = torch.randn((bl.count, 4), device="cuda", generator=gen)
                vals[:, 0] = vals[:, 0].clamp(-1, 1)
                if dims == 2:
                    vals[:, 3] = 0
                if case == "identical":
                    vals[:, 0] = canon.view(-1)[idx]
                flat[idx] = vals
            l_d, w_d = torch.empty_like(live), torch.empty(shape + (dims,), device="cuda")
            raw_d = dev.state_finalize(state, canon, grid, l_d, None, w_d, 0.3, True).cpu().numpy()
            l_l, w_l = live.clone(), torch.zeros(shape + (dims,), device="cuda")
            raw_l = dev.state_finalize_listed(state, canon, grid, lists, unlisted, l_l, w_l, 0.3, True).cpu().numpy()
            assert torch.equal(l_l, l_d) and torch.equal(w_l, w_d), (shape, case)
            for got, want in ((raw_l[:8], raw_d[:8]), (raw_l[8:], raw_d[8:])):
                assert np.array_equal(got[[0, 1, 2, 5]], want[[0, 1, 2, 5]]), (shape, case, got, want)
                assert np.allclose(got[3:5], want[3:5], rtol=1e-12, atol=0.0), (shape, case, got, want)
                assert np.array_equal(got[6:], want[6:]), (shape, case, got, want)
            # fields only
            l_l, w_l = live.clone(), torch.zeros(shape + (dims,), device="cuda")
            assert dev.state_finalize_listed(state, canon, grid, lists, unlisted, l_l, w_l) is None
            assert torch.equal(l_l, l_d) and torch.equal(w_l, w_d)


# ------------------------------------------------------------ full-size, size-independent properties
def test_state_of_4gib_and_more_takes_the_same_path(lsf):
    """a float4 state of 4 GiB and more (656^3 voxels: 4.5 GB): the list kernel's buffer-load neighbourhood addresses
    relative to each wave's first voxel there.  The same 640^3 sphere pair, padded with truncated voxels to 656^3, must
    come out bit-identical to the 640^3 run (4.19 GB of state: plain 32-bit offsets) inside the common region."""
    from levelsetfusion_python_amd import device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n, big = 640, 656
    canon, live0 = sphere_pair(n, 3, "cuda")
    assert float(live0[-1].min()) == 1.0 and float(canon[:, :, -1].min()) == 1.0  # the band stays clear of the faces
    kw = dict(compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
              smoothing_term_method=lsf.SmoothingTermMethod.KILLING, maximum_warp_length_lower_threshold=0.0,
              max_iterations=3, min_iterations=3)
    small = lsf.SlavchevaOptimizer3d(field_size=n, **kw)
    live_small = live0.clone()
    small.optimize(live_small, canon)
    warp_small = small.warp_field
    assert dev.n_voxels(dev.make_grid((n, n, n))) * 16 < 2 ** 32 <= dev.n_voxels(dev.make_grid((big, big, big))) * 16
    canon_big = torch.ones((big, big, big), device="cuda")
    live_big = torch.ones((big, big, big), device="cuda")
    canon_big[:n, :n, :n] = canon
    live_big[:n, :n, :n] = live0
    del canon, live0
    wide = lsf.SlavchevaOptimizer3d(field_size=big, **kw)
    wide.optimize(live_big, canon_big)
    assert torch.equal(live_big[:n, :n, :n], live_small)
    assert torch.equal(wide.warp_field[:n, :n, :n], warp_small)
    assert float(wide.warp_field.abs().max()) == float(warp_small.abs().max()) > 0
    assert wide.log.max_warps == small.log.max_warps
    # the arg-max locations are (x, y, z) coordinates: the same in both volumes
    assert wide.log.max_warp_locations == small.log.max_warp_locations
    assert np.allclose(wide.log.data_energies, small.log.data_energies, rtol=1e-12)


def test_full_size_2d_embedding_256(lsf):
    """BASELINE size 256^3: a z-constant volume must reproduce the 2-D result (computed by the ORACLE at 256^2)
    on interior slices, bit for bit, with w == 0 -- KillingFusion-style and hierarchical."""
    n = 256
    c2, l2 = O.sphere_pair(n, d=2)
    c3 = torch.from_numpy(c2).cuda()[None].repeat(n, 1, 1).contiguous()
    l3 = torch.from_numpy(l2).cuda()[None].repeat(n, 1, 1).contiguous()
    common = dict(maximum_warp_length_lower_threshold=0.0, max_iterations=3, min_iterations=3)
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING, **common)
    live = l3.clone()
    opt.optimize(live, c3)
    o = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
                          **common)
    live_ref = l2.copy()
    o.optimize(live_ref, c2)
    for z in (8, n // 2, n - 9):
        assert maxdiff(live[z].cpu().numpy(), live_ref) == EXACT
        assert maxdiff(opt.warp_field[z][..., :2].cpu().numpy(), o.warp_field) == EXACT
        assert float(opt.warp_field[z][..., 2].abs().max()) == 0.0
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.2,
              maximum_iteration_count=3, maximum_warp_update_threshold=0.0)
    warp3 = lsf.HierarchicalOptimizer3d(**kw).optimize(c3, l3)
    warp2 = O.HierarchicalOracle(**kw).optimize(c2, l2)
    assert maxdiff(warp3[n // 2][..., :2].cpu().numpy(), warp2) == EXACT
    assert float(warp3[n // 2][..., 2].abs().max()) == 0.0


def test_full_size_fixed_point_and_slab_invariance_256(lsf):
    """(i) live == canonical is a fixed point: zero warp, live unchanged, the loop stops after min_iterations;
    (ii) processing the 256^3 volume as two z-slabs with a 2-slice halo (kernel z-range arguments) gives exactly
    the result of the single launch."""
    from levelsetfusion_python_amd import _lib, device as dev
    n = 256
    c, l = O.sphere_pair(n, d=3)
    ct, lt = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=False,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                   maximum_warp_length_lower_threshold=0.0, max_iterations=10)
    live = ct.clone()
    opt.optimize(live, ct)
    assert len(opt.log.max_warps) == 1 and opt.log.max_warps[0] == 0.0
    assert torch.equal(live, ct) and float(opt.warp_field.abs().max()) == 0.0
    # (ii) one fused iteration, full volume vs two slabs
    eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
    warp_prev = (0.3 * torch.randn((3, n, n, n), device="cuda", generator=torch.Generator("cuda").manual_seed(3)))
    rec = dev.new_records(3, "cuda")
    s_in = dev.state_pack(lt, warp_prev, dev.make_grid(lt.shape), copies=1)[0]
    full = torch.empty_like(s_in)
    dev.slavcheva_state_iteration(s_in, ct, full, dev.make_grid(lt.shape), eng.params, None, rec, 0)
    h, half = 2, n // 2
    pieces = []
    for k, (a, b, zb, ze) in enumerate(((0, half + h, 0, half), (half - h, n, h, h + half))):
        ss, cs = s_in[a:b].contiguous(), ct[a:b].contiguous()
        out = torch.zeros_like(ss)
        dev.slavcheva_state_iteration(ss, cs, out, dev.make_grid(cs.shape, zb, ze, a), eng.params, None, rec, 1 + k)
        pieces.append(out[zb:ze])
    assert torch.equal(torch.cat(pieces, 0), full)
    d = dev.decode_records(rec.cpu().numpy())
    assert d["max_value"][0] == max(d["max_value"][1], d["max_value"][2])
    assert d["argmax"][0] == (d["argmax"][1] if d["max_value"][1] >= d["max_value"][2] else d["argmax"][2])
    assert np.isclose(d["data_energy"][0], d["data_energy"][1] + d["data_energy"][2], rtol=1e-10)


def test_full_size_config2_hierarchical_2d_512(lsf):
    """BASELINE config 2 at full size: 2-D 512 x 512 circle pair, HierarchicalOptimizer2d with a 3-level pyramid
    (maximum_chunk_size 4: 128 / 256 / 512), Tikhonov smoothing, no gradient kernel -- bit-identical to the oracle;
    levels run through the HIP-graph replay path"""
    canon, live = O.sphere_pair(512, 2)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=10, maximum_warp_update_threshold=0.0, tikhonov_strength=0.2)
    opt = lsf.HierarchicalOptimizer2d(**kw)
    warp = opt.optimize(canon, live)
    o = O.HierarchicalOracle(**kw)
    want = o.optimize(canon, live)
    assert warp.shape == (512, 512, 2) and opt.get_per_level_iteration_counts() == [10, 10, 10]
    assert maxdiff(warp, want) == EXACT
    for mine, theirs in zip(opt.get_per_level_maximum_updates(), o.per_level_max_updates):
        assert np.array_equal(np.float32(mine), np.float32(theirs))
    assert float(np.abs(want).max()) > 0.05  # the optimisation moved


def test_full_size_config3_hierarchical_3d_128_with_sobolev_kernel(lsf):
    """BASELINE config 3 at full size: 3-D 128^3 sphere pair, HierarchicalOptimizer3d, chunk 8 (4 levels 16 ... 128),
    Tikhonov + the 7-tap Sobolev gradient kernel (register-window filter passes) -- bit-identical to the oracle"""
    n = 128
    canon, live = O.sphere_pair(n, 3)
    k7 = lsf.generate_1d_sobolev_kernel(7, 0.1)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=True, maximum_chunk_size=8, rate=0.1,
              maximum_iteration_count=2, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05, kernel=k7)
    opt = lsf.HierarchicalOptimizer3d(**kw)
    warp = opt.optimize(torch.from_numpy(canon).cuda(), torch.from_numpy(live).cuda())
    o = O.HierarchicalOracle(**kw)
    want = o.optimize(canon, live)
    assert tuple(warp.shape) == (n, n, n, 3) and opt.get_per_level_iteration_counts() == [2, 2, 2, 2]
    assert maxdiff(warp.cpu().numpy(), want) == EXACT
    assert float(np.abs(want).max()) > 1e-3
    # the same with the three filter passes of every level in one launch (lsf_convolve_xyz; by default only levels of
    # 2^23 voxels and more take it)
    fused = lsf.HierarchicalOptimizer3d(engine_options=dict(fused_filter_min_voxels=0), **kw)
    warp_fused = fused.optimize(torch.from_numpy(canon).cuda(), torch.from_numpy(live).cuda())
    assert torch.equal(warp_fused, warp)
