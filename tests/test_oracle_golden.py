"""Pins the CPU oracle (oracle/lsf_oracle.py) against
  (i)  the reference tests' own known-answer literals (tests/golden/ref_test_literals.npz), and
  (ii) outputs of the reference itself, imported in the build container (tests/golden/make_golden.py).
Tolerances are absolute and written out; the north-star bound is 1e-5, the reference's own C++-vs-Python
bound is atol=10e-6 (tests/test_hierarchical_optimizer2d.py:68)."""
import numpy as np
import pytest

from oracle import lsf_oracle as O

ATOL = 1e-5       # north-star tolerance
TIGHT = 2.5e-6    # what actually holds against the reference's float32 paths


def maxdiff(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


# ------------------------------------------------------------------------------------ leaf kernels (ii)
def test_warp_functions_match_reference(ref_leaf):
    L = ref_leaf
    f, w = L["warp.field"], L["warp.warp"]
    assert maxdiff(O.warp_field(f, w), L["warp.warp_field"]) == 0.0
    assert maxdiff(O.warp_field_replacement(f, w, 0.0), L["warp.warp_field_replacement0"]) == 0.0
    assert maxdiff(O.warp_field_replacement(f, w, -0.5), L["warp.warp_field_replacement_m05"]) == 0.0


@pytest.mark.parametrize("tag,flags", [("000", (False, False, False)), ("100", (True, False, False)),
                                       ("010", (False, True, False)), ("001", (False, False, True)),
                                       ("111", (True, True, True))])
def test_warp_field_advanced_matches_reference(ref_leaf, tag, flags):
    L = ref_leaf
    w = L["warp.small_warp"].copy()
    g = (w * 10).astype(np.float32)
    new_live = O.warp_field_advanced(L["warp.canonical"], L["warp.field"].copy(), w, g, *flags)
    assert maxdiff(new_live, L["warp.advanced_%s.live" % tag]) == 0.0
    assert maxdiff(w, L["warp.advanced_%s.warp" % tag]) == 0.0
    assert maxdiff(g, L["warp.advanced_%s.gradient" % tag]) == 0.0


def test_gradient_and_pyramid_match_reference(ref_leaf):
    L = ref_leaf
    gx, gy = O.gradient(L["grad.field"])
    assert maxdiff(gx, L["grad.gx"]) == 0.0 and maxdiff(gy, L["grad.gy"]) == 0.0
    for chunk, n in ((4, 3), (8, 4)):
        levels = O.pyramid(L["grad.field"], chunk)
        assert len(levels) == n
        for i, lvl in enumerate(levels):
            assert maxdiff(lvl, L["pyramid.chunk%d.level%d" % (chunk, i)]) == 0.0


def test_pyramid_errors():
    with pytest.raises(ValueError):
        O.pyramid(np.zeros((12, 16), np.float32), 4)
    with pytest.raises(ValueError):
        O.pyramid(np.zeros((16, 16), np.float32), 3)
    with pytest.raises(ValueError):
        O.pyramid(np.zeros((8, 8), np.float32), 8)


def test_sobolev_kernels_match_reference(ref_leaf):
    for s, lam, k in ((3, 0.1, "k3"), (7, 0.1, "k7"), (9, 0.15, "k9")):
        assert maxdiff(O.generate_1d_sobolev_kernel(s, lam), ref_leaf["sobolev." + k]) <= 1e-7
    # tests/test_convolution.py:61 quotes the (3, 0.1) filter
    assert np.allclose(O.generate_1d_sobolev_kernel(3, 0.1), [0.06742075, 0.99544406, 0.06742075], atol=1e-7)


def test_convolution_matches_reference(ref_leaf):
    L = ref_leaf
    vf2, vf3 = L["conv.vf2"], L["conv.vf3"]
    assert maxdiff(O.convolve_with_kernel(vf2.copy(), L["sobolev.hardcoded7"]), L["conv.vf2_k7f64"]) == 0.0
    assert maxdiff(O.convolve_with_kernel(vf3.copy(), L["sobolev.hardcoded7"]), L["conv.vf3_k7f64"]) == 0.0
    assert maxdiff(O.convolve_with_kernel(vf2.copy(), L["sobolev.k7"]), L["conv.vf2_k7f32"]) <= TIGHT
    assert maxdiff(O.convolve_with_kernel(vf2.copy(), np.array([0.5, 0.2, -0.1, 0.05, 0.3])),
                   L["conv.vf2_asym"]) <= TIGHT
    assert maxdiff(O.convolve_with_kernel(vf3.copy(), np.array([0.5, 0.2, -0.1])), L["conv.vf3_asym"]) <= TIGHT
    assert maxdiff(O.convolve_with_kernel_preserve_zeros(vf2.copy(), L["sobolev.k3"]), L["conv.vf2_pz_k3"]) <= TIGHT
    assert maxdiff(O.convolve_with_kernel_preserve_zeros(vf2.copy(), L["sobolev.hardcoded7"]),
                   L["conv.vf2_pz_k7"]) == 0.0


def test_even_length_kernels_follow_numpy_same():
    """np.convolve(line, k, 'same') (math_utils/convolution.py:78-83) cuts the full convolution at (len(k) - 1) // 2:
    for an even-length, asymmetric kernel that is one sample away from len(k) // 2"""
    rng = np.random.default_rng(5)
    for kern in (np.array([0.4, -0.2, 0.7, 0.1]), np.array([1.0, 2.0]), np.array([0.3, 0.1, -0.5, 0.25, 0.6, -0.05])):
        for shape, order in (((9, 11, 2), (0, 1)), ((6, 7, 8, 3), (2, 1, 0))):
            vf = rng.standard_normal(shape).astype(np.float32)
            want = vf.astype(np.float64)
            for axis in order:  # the reference's pass order, every line through numpy itself
                want = np.apply_along_axis(lambda line: np.convolve(line, kern, "same"), axis,
                                           want.astype(np.float32).astype(np.float64))
            assert maxdiff(O.convolve_with_kernel(vf.copy(), kern), want.astype(np.float32)) <= TIGHT


def test_slavcheva_terms_match_reference(ref_leaf):
    L = ref_leaf
    live, canon, warp = L["terms.live"], L["terms.canonical"], L["terms.warp"]
    gd, diff = O.data_term_gradient(live, canon)
    assert maxdiff(gd, L["terms.data_vectorized"]) == 0.0
    assert maxdiff(gd, L["terms.data_basic"]) == 0.0
    band = ~(O.is_truncated(live) & O.is_truncated(canon))
    assert abs(0.5 * float((diff.astype(np.float64)[band] ** 2).sum()) - float(L["terms.data_energy"])) < 1e-6
    assert maxdiff(O.tikhonov_gradient(warp), L["terms.tikhonov_vectorized"]) == 0.0
    assert maxdiff(O.tikhonov_gradient(warp), L["terms.tikhonov_direct"]) <= TIGHT
    assert maxdiff(O.tikhonov_energy_direct(warp), L["terms.tikhonov_direct_energy"]) <= TIGHT
    assert abs(O.smoothing_energy_vectorized(warp, band) - float(L["terms.smoothing_energy_vectorized"])) < 1e-4
    kg, ke = O.killing_gradient(warp, 0.1)
    assert maxdiff(kg, L["terms.killing"]) == 0.0
    assert maxdiff(ke, L["terms.killing_energy"]) <= TIGHT
    lg, le = O.level_set_gradient(live)
    assert maxdiff(lg, L["terms.level_set"]) == 0.0
    assert maxdiff(le, L["terms.level_set_energy"]) <= 1e-5
    gf, _ = O.data_term_gradient(L["terms.steep_live"], canon, O.THRESHOLDED_FDM)
    assert maxdiff(gf, L["terms.data_fdm"]) <= TIGHT


# ------------------------------------------------------------------------- reference test literals (i)
def test_reference_known_answers_field_warping(ref_literals):
    T = ref_literals
    assert maxdiff(O.warp_field(T["hierarchical_data.field_A_16x16"], T["hierarchical_data.warp_field_A_16x16"]),
                   T["hierarchical_data.fA_resampled_with_wfA"]) <= TIGHT
    assert maxdiff(O.warp_field_replacement(T["hierarchical_data.field_B_16x16"],
                                            T["hierarchical_data.warp_field_B_16x16"], 0.0),
                   T["hierarchical_data.fB_resampled_with_wfB_replacement"]) <= TIGHT
    # the five warp_field_advanced cases (tests/test_field_warping.py:25-262); flags per case
    flags = {"01": (False, False, False), "02": (True, False, True), "03": (False, False, False),
             "04": (False, False, False), "05": (False, False, False)}
    for case, fl in flags.items():
        p = "field_warping.test_warp_field_advanced%s." % case
        warp = np.stack((T[p + "u_vectors"], T[p + "v_vectors"]), axis=2)
        grad = (warp * 10).astype(np.float32)
        new_live = O.warp_field_advanced(T[p + "canonical_field"], T[p + "warped_live_template"].copy(), warp, grad, *fl)
        assert np.allclose(new_live, T[p + "expected_new_warped_live_field"], atol=TIGHT), case
        if p + "expected_u_vectors" in T.files:  # cases 04/05 only check the live field
            assert np.allclose(warp[..., 0], T[p + "expected_u_vectors"], atol=TIGHT), case
            assert np.allclose(warp[..., 1], T[p + "expected_v_vectors"], atol=TIGHT), case


def test_reference_known_answers_convolution(ref_literals):
    T = ref_literals
    field = np.array([1, 4, 7, 2, 5, 8, 3, 6, 9], dtype=np.float32).reshape(3, 3)
    vf = np.dstack([field] * 2)
    O.convolve_with_kernel_preserve_zeros(vf, np.flip(np.array([1, 2, 3])))
    assert np.allclose(vf[..., 0], [[85, 168, 99], [124, 228, 132], [67, 120, 69]])
    p = "convolution.test_convolve_with_kernel_preserve_zeros02."
    vf = T[p + "vector_field"].copy()
    O.convolve_with_kernel_preserve_zeros(vf, np.flip(T[p + "kernel"]))
    assert np.allclose(vf, T[p + "expected_output"], rtol=0.0, atol=1e-6)
    p = "convolution.test_convolve_with_kernel_2d."
    vf = T[p + "vector_field"].copy()
    O.convolve_with_kernel(vf, np.flip(T[p + "kernel"]))
    assert np.allclose(vf, T[p + "expected_output"], atol=1e-6)
    vf3 = np.arange(1.0, 241.0).reshape(5, 4, 4, 3).astype(np.float32)
    O.convolve_with_kernel(vf3, np.array([3.0, 2.0, 1.0]))
    assert np.allclose(vf3, T["convolution_data.convolved_3d_vector_field"])


def test_reference_known_answers_pyramid():
    tile = np.array([[1, 2, 5, 6, -1, -2, -5, -6], [3, 4, 7, 8, -3, -4, -7, -8],
                     [-1, -2, -5, -6, 1, 2, 5, 6], [-3, -4, -7, -8, 3, 4, 7, 8],
                     [1, 2, 5, 6, 5, 5, 5, 5], [3, 4, 7, 8, 5, 5, 5, 5],
                     [-1, -2, -5, -6, 5, 5, 5, 5], [-3, -4, -7, -8, 5, 5, 5, 5]], dtype=np.float32)
    levels = O.pyramid(np.tile(tile, (16, 16)))
    assert [l.shape for l in levels] == [(16, 16), (32, 32), (64, 64), (128, 128)]
    assert levels[2][0, 0] == tile[0:2, 0:2].mean() and levels[2][1, 1] == tile[2:4, 2:4].mean()
    assert levels[2][0, 2] == -tile[0:2, 0:2].mean()
    assert levels[1][1, 1] == 5.0 and levels[0][0, 0] == 5.0 / 4


def test_reference_golden_hierarchical_16x16(ref_literals):
    """tests/test_hierarchical_optimizer2d.py:39-54 -- 4 levels, 1+100+100+100 iterations."""
    T = ref_literals
    o = O.HierarchicalOracle(rate=0.2, data_term_amplifier=1.0, maximum_warp_update_threshold=0.001,
                             maximum_iteration_count=100, tikhonov_term_enabled=False, kernel=None)
    warp = o.optimize(T["hierarchical_data.canonical_field"], T["hierarchical_data.live_field"])
    assert o.per_level_iteration_counts == [1, 100, 100, 100]
    assert maxdiff(warp, T["hierarchical_data.warp_field"]) <= TIGHT
    assert maxdiff(O.warp_field(T["hierarchical_data.live_field"], warp),
                   T["hierarchical_data.final_live_field"]) <= TIGHT


def test_reference_golden_slavcheva_4x4_and_report(ref_literals, ref_slavcheva):
    """tests/test_slavcheva_optimizer.py:64-149."""
    T = ref_literals
    live0 = T["slavcheva.test_nonrigid_optimization01.live_field_template"]
    canon = T["slavcheva.test_nonrigid_optimization01.canonical_field"]
    k3 = ref_slavcheva["kernel3"]
    for method in (O.VECTORIZED, O.DIRECT):
        for n_it, key in ((1, "slavcheva.test_nonrigid_optimization01.expected_live_field_out"),
                          (2, "slavcheva.test_nonrigid_optimization02.expected_live_field_out")):
            o = O.SlavchevaOracle(compute_method=method, sobolev_smoothing_enabled=True,
                                  maximum_warp_length_lower_threshold=0.05, max_iterations=n_it, sobolev_kernel=k3)
            live = live0.copy()
            assert o.optimize(live, canon) is live
            assert o.iteration_count == n_it
            assert maxdiff(live, T[key]) <= TIGHT
    ws = O.warp_delta_statistics(o.warp_field, canon, live, 0.05, 10000)
    exp = T["slavcheva.report.warp_stats"]
    got = [ws["ratio_above_min_threshold"], ws["length_min"], ws["length_max"], ws["length_mean"],
           ws["length_standard_deviation"], *ws["longest_warp_location"]]
    assert np.allclose(got, exp, atol=1e-6)
    ds = O.tsdf_difference_statistics(canon, live)
    got = [ds["difference_min"], ds["difference_max"], ds["difference_mean"],
           ds["difference_standard_deviation"], *ds["biggest_difference_location"]]
    assert np.allclose(got, T["slavcheva.report.tsdf_stats"], atol=1e-6)


# ----------------------------------------------------------------------- whole-optimizer runs (ii)
def _hier_cases(H):
    for key in H.files:
        if key.endswith(".final_warp") and "threshold" not in key:
            case, tik, ker, chunk, _ = key.split(".")
            yield key, case, tik == "tik1", ker == "ker1", int(chunk[5:])


def test_hierarchical_runs_match_reference(ref_hierarchical, ref_literals):
    H, T = ref_hierarchical, ref_literals
    cases = {"g16": (T["hierarchical_data.canonical_field"], T["hierarchical_data.live_field"]),
             "c64": (H["c64.canonical"], H["c64.live"])}
    n = 0
    for key, case, tik, ker, chunk in _hier_cases(H):
        o = O.HierarchicalOracle(tikhonov_term_enabled=tik, gradient_kernel_enabled=ker, maximum_chunk_size=chunk,
                                 rate=0.2, maximum_iteration_count=4, maximum_warp_update_threshold=0.0,
                                 tikhonov_strength=0.2, kernel=H["kernel7"] if ker else None)
        per = {}
        o.iteration_hook = lambda level, it, warp, g, m: per.__setitem__((level, it), warp.copy())
        warp = o.optimize(*cases[case])
        assert maxdiff(warp, H[key]) <= (1e-7 if ker else 0.0), key
        for (level, it), w in per.items():
            k = "%s.L%d.it%d.warp" % (key[:-len(".final_warp")], level, it)
            if k in H.files:
                assert maxdiff(w, H[k]) <= (1e-7 if ker else 0.0), k
        n += 1
    assert n >= 10
    o = O.HierarchicalOracle(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8,
                             rate=0.1, maximum_iteration_count=40, maximum_warp_update_threshold=0.01)
    warp = o.optimize(*cases["c64"])
    assert o.per_level_iteration_counts == list(H["c64.threshold_run.iteration_counts"])
    assert maxdiff(warp, H["c64.threshold_run.final_warp"]) == 0.0


SLAVCHEVA_CONFIGS = {
    "sobolev_vec": dict(compute_method=O.VECTORIZED, sobolev_smoothing_enabled=True, kernel="kernel7"),
    "sobolev_direct": dict(compute_method=O.DIRECT, sobolev_smoothing_enabled=True, kernel="kernel3"),
    "killing": dict(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING),
    "tikhonov_direct": dict(compute_method=O.DIRECT),
    "fdm_direct": dict(compute_method=O.DIRECT, data_term_method=O.THRESHOLDED_FDM),
}


@pytest.mark.parametrize("size_tag,n_it", [("ortho32", 4), ("ortho64", 3)])
@pytest.mark.parametrize("name", sorted(SLAVCHEVA_CONFIGS))
def test_slavcheva_runs_match_reference(ref_slavcheva, size_tag, n_it, name):
    S = ref_slavcheva
    kw = dict(SLAVCHEVA_CONFIGS[name])
    kernel = kw.pop("kernel", None)
    o = O.SlavchevaOracle(maximum_warp_length_lower_threshold=0.0, max_iterations=n_it, min_iterations=n_it,
                          sobolev_kernel=S[kernel] if kernel else None, **kw)
    per = {}
    o.iteration_hook = lambda it, live, warp, g, en, mw, at: per.__setitem__(it, (warp.copy(), g.copy(), live.copy()))
    live = S[size_tag + ".live"].copy()
    o.optimize(live, S[size_tag + ".canonical"])
    tag = size_tag + "." + name
    assert maxdiff(live, S[tag + ".final_live"]) <= TIGHT
    assert maxdiff(o.warp_field, S[tag + ".final_warp"]) <= TIGHT
    assert maxdiff(o.gradient_field, S[tag + ".final_gradient"]) <= ATOL
    assert maxdiff(o.log["max_warps"], S[tag + ".max_warps"]) <= TIGHT
    for mine, theirs in ((o.log["data_energies"], S[tag + ".data_energies"]),
                         (o.log["smoothing_energies"], S[tag + ".smoothing_energies"]),
                         (o.log["level_set_energies"], S[tag + ".level_set_energies"])):
        assert np.allclose(mine, theirs, rtol=1e-5, atol=1e-7)
    for it, (w, g, l) in per.items():
        if "%s.it%d.warp" % (tag, it) in S.files:
            assert maxdiff(w, S["%s.it%d.warp" % (tag, it)]) <= TIGHT
            assert maxdiff(g, S["%s.it%d.gradient" % (tag, it)]) <= ATOL
            assert maxdiff(l, S["%s.it%d.live" % (tag, it)]) <= TIGHT


@pytest.mark.parametrize("name", ["sobolev_vec", "killing"])
def test_config1_one_hundred_iterations_match_reference(ref_slavcheva, ref_config1, name):
    """BASELINE config 1 at its full length (64 x 64 orthographic pair, 100 fixed iterations) against the reference's own
    run (tests/golden/make_golden.py config1): KillingFusion (DIRECT) bit for bit in every snapshot -- through the snap
    discontinuity of warp_field_advanced and updates of 4 to 6 voxels per iteration --, SobolevFusion (VECTORIZED, 7-tap
    filter: numpy accumulates its float32 convolution in an unspecified order) within 5e-6"""
    S, C = ref_slavcheva, ref_config1
    kw = dict(SLAVCHEVA_CONFIGS[name])
    kernel = kw.pop("kernel", None)
    o = O.SlavchevaOracle(maximum_warp_length_lower_threshold=0.0, max_iterations=100, min_iterations=100,
                          sobolev_kernel=S[kernel] if kernel else None, **kw)
    per = {}
    o.iteration_hook = lambda it, live, warp, g, en, mw, at: per.__setitem__(it, (warp.copy(), live.copy())) \
        if it in (9, 24, 49, 99) else None
    live = S["ortho64.live"].copy()
    o.optimize(live, S["ortho64.canonical"])
    tag = "ortho64.%s.100" % name
    tol = 0.0 if name == "killing" else 5e-6
    assert maxdiff(live, C[tag + ".final_live"]) <= tol
    assert maxdiff(o.gradient_field, C[tag + ".final_gradient"]) <= 10.0 * tol  # warp = -gradient * rate, rate 0.1
    assert maxdiff(o.log["max_warps"], C[tag + ".max_warps"]) <= tol
    assert sorted(per) == [9, 24, 49, 99]
    for it, (w, l) in per.items():
        assert maxdiff(w, C["%s.it%d.warp" % (tag, it)]) <= tol
        assert maxdiff(l, C["%s.it%d.live" % (tag, it)]) <= tol
    for mine, theirs in ((o.log["data_energies"], C[tag + ".data_energies"]),
                         (o.log["smoothing_energies"], C[tag + ".smoothing_energies"]),
                         (o.log["level_set_energies"], C[tag + ".level_set_energies"])):
        assert np.allclose(mine, theirs, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------- 3-D rules: 2-D embedding (pins A.12)
def test_3d_embedding_of_2d_is_exact():
    rng = np.random.default_rng(5)
    n, nz = 16, 6
    f2 = np.clip(0.1 * (np.arange(n)[:, None] - 7.3) + 0.2 * np.sin(np.arange(n)[None, :] * 0.5)
                 + 0.05 * rng.standard_normal((n, n)), -1, 1).astype(np.float32)
    c2 = np.clip(f2 + 0.1 * rng.standard_normal((n, n)), -1, 1).astype(np.float32)
    w2 = (0.6 * rng.standard_normal((n, n, 2))).astype(np.float32)
    f3, c3 = np.repeat(f2[None], nz, 0), np.repeat(c2[None], nz, 0)
    w3 = np.concatenate((np.repeat(w2[None], nz, 0), np.zeros((nz, n, n, 1), np.float32)), axis=-1)
    mid = slice(1, nz - 1)
    assert maxdiff(O.warp_field(f3, w3)[mid], np.repeat(O.warp_field(f2, w2)[None], nz - 2, 0)) == 0.0
    assert maxdiff(O.restrict_mean(f3)[1], O.restrict_mean(f2)) == 0.0
    assert maxdiff(O.laplace_replicate(f3)[2], O.laplace_replicate(f2)) == 0.0
    k3, _ = O.killing_gradient(w3, 0.1)
    k2, _ = O.killing_gradient(w2, 0.1)
    assert maxdiff(k3[2][..., :2], k2) == 0.0 and maxdiff(k3[2][..., 2], 0) == 0.0
    l3, _ = O.level_set_gradient(f3)
    l2, _ = O.level_set_gradient(f2)
    assert maxdiff(l3[2][..., :2], l2) == 0.0 and maxdiff(l3[2][..., 2], 0) == 0.0
    # full optimizers: hierarchical (no kernel) and KillingFusion-style, interior slices
    n, nz = 32, 32
    c2, l2 = O.sphere_pair(n, d=2)
    c3, l3 = np.repeat(c2[None], nz, 0), np.repeat(l2[None], nz, 0)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.2,
              maximum_iteration_count=3, maximum_warp_update_threshold=0.0)
    wa = O.HierarchicalOracle(**kw).optimize(c2, l2)
    wb = O.HierarchicalOracle(**kw).optimize(c3, l3)
    assert maxdiff(wb[nz // 2][..., :2], wa) == 0.0 and maxdiff(wb[nz // 2][..., 2], 0) == 0.0
    kw = dict(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
              maximum_warp_length_lower_threshold=0.0, max_iterations=3, min_iterations=3)
    la, lb = l2.copy(), l3.copy()
    oa, ob = O.SlavchevaOracle(**kw), O.SlavchevaOracle(**kw)
    oa.optimize(la, c2)
    ob.optimize(lb, c3)
    assert maxdiff(lb[nz // 2], la) == 0.0
    assert maxdiff(ob.warp_field[nz // 2][..., :2], oa.warp_field) == 0.0


# ------------------------------------------------------------------------------------ a21 TSDF generation
def _tsdf_cases(G):
    K = G["intrinsics"]
    d0, d1 = O.synthetic_depth_image(), O.synthetic_depth_image(shift_px=2.0, nearer_m=0.008)
    for d, key in ((d0, "depth0.checksum"), (d1, "depth1.checksum")):
        assert [int(d.astype(np.int64).sum()), int((d.astype(np.int64) ** 2).sum())] == list(G[key])
    cases = []
    for tag, d in (("d0", d0), ("d1", d1)):
        for y in (200, 240):
            cases.append(("%s.row%d.n32" % (tag, y), d, K, dict(field_shape=(32, 32), image_y_coordinate=y,
                                                                array_offset=(-16, -16, 234))))
        cases.append(("%s.vol16" % tag, d, K, dict(field_shape=(16, 16, 16), array_offset=(-8, -8, 240))))
    cases.append(("d0.vol12.extrinsic", d0, K, dict(field_shape=(12, 12, 12), array_offset=(-6, -6, 244),
                                                    camera_extrinsic_matrix=G["extrinsic"])))
    cases.append(("d0.vol12.k64", d0, K.astype(np.float64), dict(field_shape=(12, 12, 12),
                                                                 array_offset=(-6, -6, 244))))
    cases.append(("d0.row240.n32.default0", d0, K, dict(field_shape=(32, 32), image_y_coordinate=240, default_value=0,
                                                        array_offset=(-16, -16, 234), narrow_band_width_voxels=10)))
    return cases


def test_tsdf_nearest_matches_reference(ref_tsdf):
    n_band = 0
    for key, depth, K, kw in _tsdf_cases(ref_tsdf):
        got = O.tsdf_nearest(depth, K, 0.001, **kw)
        tol = 2.5e-6 if "extrinsic" in key else 0.0  # BLAS matvec order for a general extrinsic matrix
        assert maxdiff(got, ref_tsdf[key]) <= tol, key
        n_band += int((np.abs(ref_tsdf[key]) < 1).sum())
    assert n_band > 3000


def _tsdf_bilinear_cases(G):
    """(key, depth, intrinsics, tsdf_space, row, kwargs of oracle.tsdf_bilinear) for every bilinear fixture"""
    K, E = G["intrinsics"], G["extrinsic"]
    d0, d1 = O.synthetic_depth_image(), O.synthetic_depth_image(shift_px=2.0, nearer_m=0.008)
    cases = []
    for key in sorted(k for k in G.files if "bilinear" in k):
        parts = key.split(".")
        kw = dict(array_offset=(-16, -16, 234))
        if "border" in key:  # the right-most voxels project past the image border
            kw = dict(array_offset=(90, -16, 234), default_value=0, narrow_band_width_voxels=10)
        if "extrinsic" in key:
            kw["camera_extrinsic_matrix"] = E
        cases.append((key, d0 if parts[0] == "d0" else d1, K.astype(np.float64) if "k64" in key else K,
                      parts[1] == "bilinear_tsdf", int(parts[2][3:]), kw))
    assert len(cases) == 10
    return cases


def test_tsdf_bilinear_matches_reference(ref_tsdf):
    """tsdf/generation.py:18-128 (bilinear image space / TSDF space) run by the reference itself"""
    for key, depth, K, tsdf_space, row, kw in _tsdf_bilinear_cases(ref_tsdf):
        got = O.tsdf_bilinear(depth, K, 0.001, 32, row, tsdf_space, **kw)
        assert maxdiff(got, ref_tsdf[key]) <= (2.5e-6 if "extrinsic" in key else 0.0), key
        assert int((np.abs(ref_tsdf[key]) < 1).sum()) > 150, key
    border = ref_tsdf["d0.bilinear_image.row240.n32.border"]
    assert int((border == 0).sum()) > 100  # voxels whose projection left the image kept the default value


# ------------------------------------------------------------------------------------ a21 EWA TSDF generation
def _banded_image(rows, band):
    img = np.full((480, 640), np.iinfo(np.uint16).max, dtype=np.uint16)
    img[int(rows[0]):int(rows[1])] = band
    return img


def _ewa_cases(G):
    """(key, depth image, kwargs of oracle.tsdf_ewa, expected-literal key of the reference's own test data or None)"""
    K = np.array([[700., 0., 320.], [0., 700., 240.], [0., 0., 1.]], dtype=np.float32)
    z1, z2 = _banded_image(G["rows"], G["zigzag1.rows"]), _banded_image(G["rows"], G["zigzag2.rows"])
    d0 = O.synthetic_depth_image()
    return K, [
        ("case1.image.patch", G["patch"], dict(field_shape=(16, 16), method=O.EWA_IMAGE, image_y_coordinate=1,
                                               array_offset=(94, -256, 804)), "expected.out_sdf_field01"),
        ("case2.image.zigzag2", z2, dict(field_shape=(16, 16), method=O.EWA_IMAGE, image_y_coordinate=200,
                                         array_offset=(-46, 0, 103)), "expected.out_sdf_chunk"),
        ("case3.voxel.zigzag1", z1, dict(field_shape=(16, 16), method=O.EWA_VOXEL, image_y_coordinate=200,
                                         array_offset=(-232, -256, 490), gaussian_covariance_scale=0.5),
         "expected.out_sdf_field03"),
        ("case4.inclusive.zigzag1", z1, dict(field_shape=(16, 16), method=O.EWA_VOXEL_INCLUSIVE,
                                             image_y_coordinate=200, array_offset=(-232, -256, 490),
                                             gaussian_covariance_scale=0.5), "expected.out_sdf_field04"),
        ("case5.image3d.zigzag2", z2, dict(field_shape=(16, 1, 16), method=O.EWA_IMAGE,
                                           array_offset=(-46, -8, 105)), "expected.sdf_3d_slice01"),
        ("syn.image2d", d0, dict(field_shape=(20, 20), method=O.EWA_IMAGE, image_y_coordinate=240,
                                 array_offset=(-10, -10, 236)), None),
        ("syn.voxel2d", d0, dict(field_shape=(20, 20), method=O.EWA_VOXEL, image_y_coordinate=240,
                                 array_offset=(-10, -10, 236), gaussian_covariance_scale=2.0), None),
        ("syn.inclusive2d.border", d0, dict(field_shape=(20, 20), method=O.EWA_VOXEL_INCLUSIVE, image_y_coordinate=0,
                                            array_offset=(100, -10, 236), gaussian_covariance_scale=2.0), None),
        ("syn.image3d.extrinsic", d0, dict(field_shape=(10, 6, 10), method=O.EWA_IMAGE, array_offset=(-5, -3, 238),
                                           camera_extrinsic_matrix=G["extrinsic"]), None),
    ]


def test_tsdf_ewa_matches_reference_and_its_known_answers(ref_ewa):
    K, cases = _ewa_cases(ref_ewa)
    for key, depth, kw, expected in cases:
        got = O.tsdf_ewa(depth, K, 0.001, **kw)
        assert maxdiff(got, ref_ewa[key]) <= (2.5e-6 if "extrinsic" in key else 0.0), key
        if expected is not None:  # the reference's own tolerance for these literals is atol=2e-5
            assert maxdiff(got, ref_ewa[expected]) <= 2e-5, key
        assert (np.abs(got) < 1).sum() > 10, key
