#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (read-only, /root/reference)
in the build container.  The fixtures are data only: inputs (seeded / closed-form / the reference tests'
own literal arrays re-captured as arrays) and the outputs the reference's numpy code produced for them.
No reference source text is stored.  Run:   python tests/golden/make_golden.py

Outputs
  ref_test_literals.npz   literal arrays held by the reference's own tests (known answers), by name
  ref_leaf.npz            leaf functions of SURVEY section 8a run by the reference on seeded inputs
  ref_hierarchical.npz    HierarchicalOptimizer2d runs (per-iteration warp fields)
  ref_slavcheva.npz       SlavchevaOptimizer2d runs (per-iteration live / warp / gradient / energies)
  ref_config1.npz         BASELINE config 1 at full length: the 64 x 64 orthographic pair, 100 fixed iterations
  ref_focus.npz           the "focus neighbourhood" per-voxel traces of DIRECT SlavchevaOptimizer2d runs
  ref_orthographic.npz    the hand-made orthographic 2-D pairs themselves (tsdf/generation.py:238-353) and the errors
                          the generator raises for fields that cannot hold them
"""
import ast
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstubs  # noqa: E402

_refstubs.install()
REF = _refstubs.REFERENCE_ROOT

import utils.sampling as sampling  # noqa: E402
from nonrigid_opt import field_warping as fw  # noqa: E402
from nonrigid_opt.hierarchical import hierarchical_optimizer2d as ho  # noqa: E402
from nonrigid_opt.hierarchical import hierarchical_optimization_visualizer as hov  # noqa: E402
from nonrigid_opt.hierarchical.pyramid import ScalarFieldPyramid2d  # noqa: E402
from nonrigid_opt.slavcheva import data_term as dt, smoothing_term as st  # noqa: E402
from nonrigid_opt.slavcheva import slavcheva_optimizer2d as so  # noqa: E402
from nonrigid_opt.slavcheva import slavcheva_visualizer as sviz  # noqa: E402
from nonrigid_opt.slavcheva.level_set_term import level_set_term_at_location  # noqa: E402
from nonrigid_opt.slavcheva.sobolev_filter import generate_1d_sobolev_kernel  # noqa: E402
import math_utils.convolution as mc  # noqa: E402
import math_utils.resampling as mr  # noqa: E402
import tsdf.generation as tsdf_gen  # noqa: E402
gen_mod = tsdf_gen


# ------------------------------------------------------------------------------------------------------
def extract_test_literals(rel_path, out, prefix):
    """Evaluate every `name = np.array(<literal>...)`-style assignment inside the test methods of a
    reference test file and store the resulting ARRAY under '<prefix>.<test name>.<variable>[#k]'."""
    with open(os.path.join(REF, rel_path)) as f:
        tree = ast.parse(f.read())
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name.startswith("test_")]:
            seen = {}
            for node in ast.walk(fn):
                if not (isinstance(node, ast.Assign) and len(node.targets) == 1
                        and isinstance(node.targets[0], ast.Name)):
                    continue
                names = {n.id for n in ast.walk(node.value) if isinstance(n, ast.Name)}
                if not names <= {"np"} or "np" not in names:
                    continue
                try:
                    val = eval(compile(ast.Expression(node.value), "<lit>", "eval"), {"np": np})
                except Exception:
                    continue
                if not isinstance(val, np.ndarray):
                    continue
                var = node.targets[0].id
                k = seen.get(var, 0)
                seen[var] = k + 1
                key = "%s.%s.%s%s" % (prefix, fn.name, var, "" if k == 0 else "#%d" % k)
                out[key] = val


def literals():
    out = {}
    for rel, prefix in (("tests/test_field_warping.py", "field_warping"),
                        ("tests/test_convolution.py", "convolution"),
                        ("tests/test_data_term.py", "data_term"),
                        ("tests/test_smoothing_term.py", "smoothing_term"),
                        ("tests/test_slavcheva_optimizer.py", "slavcheva"),
                        ("tests/test_field_pyramid.py", "pyramid"),
                        ("tests/test_math.py", "math")):
        extract_test_literals(rel, out, prefix)
    import tests.test_data.hierarchical_optimizer_test_data as hd
    import tests.test_data.test_data_convolution as cd
    for mod, prefix in ((hd, "hierarchical_data"), (cd, "convolution_data")):
        for name in dir(mod):
            v = getattr(mod, name)
            if isinstance(v, np.ndarray):
                out["%s.%s" % (prefix, name)] = v
    # scalar known answers of tests/test_slavcheva_optimizer.py:141-145 (convergence report), as numbers
    out["slavcheva.report.warp_stats"] = np.array([0.272727, 0.0, 0.0684823, 0.0364445, 0.0167321, 1, 2])
    out["slavcheva.report.tsdf_stats"] = np.array([0, 0.246834, 0.111843, 0.0812234, 3, 3])
    np.savez_compressed(os.path.join(HERE, "ref_test_literals.npz"), **out)
    print("ref_test_literals.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------------------------
def smooth_random_field(rng, n, scale):
    """smooth-ish TSDF-like test field in [-1, 1] with exact +-1 plateaus"""
    yy, xx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    f = 0.08 * (yy - n / 2) + 0.3 * np.sin(xx * 0.35) + scale * rng.standard_normal((n, n))
    return np.clip(f, -1.0, 1.0).astype(np.float32)


def leaf():
    rng = np.random.default_rng(20240607)
    out = {}
    n = 12
    field = smooth_random_field(rng, n, 0.05)
    canon = smooth_random_field(rng, n, 0.05)
    warp = (1.7 * rng.standard_normal((n, n, 2))).astype(np.float32)
    warp[0, 0] = (-3.2, 0.4)
    warp[n - 1, n - 1] = (2.6, 5.1)
    warp[3, 4] = (0.0, 0.0)
    warp[5, 5] = (1.0, -2.0)  # integer displacement
    out["warp.field"], out["warp.canonical"], out["warp.warp"] = field, canon, warp
    out["warp.warp_field"] = fw.warp_field(field, warp)
    out["warp.warp_field_replacement0"] = fw.warp_field_replacement(field, warp, 0.0)
    out["warp.warp_field_replacement_m05"] = fw.warp_field_replacement(field, warp, -0.5)
    small = (0.4 * rng.standard_normal((n, n, 2))).astype(np.float32)
    out["warp.small_warp"] = small
    for tag, flags in (("000", (False, False, False)), ("100", (True, False, False)),
                       ("010", (False, True, False)), ("001", (False, False, True)),
                       ("111", (True, True, True))):
        w = small.copy()
        g = (w * 10).astype(np.float32)
        with contextlib.redirect_stdout(io.StringIO()):
            new_live = fw.warp_field_advanced(canon, field.copy(), w, g, *flags)
        out["warp.advanced_%s.live" % tag] = new_live
        out["warp.advanced_%s.warp" % tag] = w
        out["warp.advanced_%s.gradient" % tag] = g

    # np.gradient + pyramid
    f32 = smooth_random_field(rng, 32, 0.1)
    gy, gx = np.gradient(f32)
    out["grad.field"], out["grad.gx"], out["grad.gy"] = f32, gx, gy
    for chunk in (4, 8):
        for i, lvl in enumerate(ScalarFieldPyramid2d(f32, chunk).levels):
            out["pyramid.chunk%d.level%d" % (chunk, i)] = lvl.astype(np.float32)

    # convolution, 2-D and 3-D, float64 7-tap (hard-coded) and float32 generated kernels
    vf2 = rng.standard_normal((16, 16, 2)).astype(np.float32)
    vf2[np.abs(vf2) < 0.3] = 0.0
    vf3 = rng.standard_normal((8, 9, 7, 3)).astype(np.float32)
    k7 = generate_1d_sobolev_kernel(size=7, strength=0.1)
    k3 = generate_1d_sobolev_kernel(size=3, strength=0.1)
    k9 = generate_1d_sobolev_kernel(size=9, strength=0.15)
    out["sobolev.k3"], out["sobolev.k7"], out["sobolev.k9"] = k3, k7, k9
    out["sobolev.hardcoded7"] = mc.sobolev_kernel_1d
    out["conv.vf2"], out["conv.vf3"] = vf2, vf3
    out["conv.vf2_k7f64"] = mc.convolve_with_kernel(vf2.copy(), mc.sobolev_kernel_1d)
    out["conv.vf2_k7f32"] = mc.convolve_with_kernel(vf2.copy(), k7)
    out["conv.vf2_asym"] = mc.convolve_with_kernel(vf2.copy(), np.array([0.5, 0.2, -0.1, 0.05, 0.3]))
    out["conv.vf3_k7f64"] = mc.convolve_with_kernel(vf3.copy(), mc.sobolev_kernel_1d)
    out["conv.vf3_asym"] = mc.convolve_with_kernel(vf3.copy(), np.array([0.5, 0.2, -0.1]))
    out["conv.vf2_pz_k3"] = mc.convolve_with_kernel_preserve_zeros(vf2.copy(), k3)
    out["conv.vf2_pz_k7"] = mc.convolve_with_kernel_preserve_zeros(vf2.copy(), mc.sobolev_kernel_1d)

    # Slavcheva leaf terms on a 12x12 case
    live = field.copy()
    live[2, 3] = 1.0
    live[2, 4] = 1.0
    live[7, 7] = -1.0
    canon2 = canon.copy()
    canon2[2, 3] = 1.0
    canon2[7, 7] = 1.0
    lgy, lgx = np.gradient(live)
    out["terms.live"], out["terms.canonical"], out["terms.warp"] = live, canon2, small
    out["terms.data_vectorized"] = dt.compute_data_term_gradient_vectorized(live, canon2, lgx, lgy)
    out["terms.data_energy"] = np.array(dt.compute_data_term_energy_contribution(live, canon2))
    out["terms.tikhonov_vectorized"] = st.compute_smoothing_term_gradient_vectorized(small)
    out["terms.smoothing_energy_vectorized"] = np.array(st.compute_smoothing_term_energy(small, live, canon2))
    steep = np.clip(live * 6.0, -1, 1).astype(np.float32)
    sgy, sgx = np.gradient(steep)
    out["terms.steep_live"] = steep
    data_fdm = np.zeros((n, n, 2), np.float32)
    data_basic = np.zeros((n, n, 2), np.float32)
    tik = np.zeros((n, n, 2), np.float32)
    tik_e = np.zeros((n, n))
    kil = np.zeros((n, n, 2), np.float32)
    kil_e = np.zeros((n, n))
    ls = np.zeros((n, n, 2), np.float32)
    ls_e = np.zeros((n, n))
    sampling.set_focus_coordinates(-5, -5)
    for y in range(n):
        for x in range(n):
            data_fdm[y, x] = dt.compute_local_data_term_gradient_thresholded_fdm(steep, canon2, x, y, sgx, sgy)[0]
            data_basic[y, x] = dt.compute_local_data_term_gradient_basic(live, canon2, x, y, lgx, lgy)[0]
            tik[y, x], tik_e[y, x] = st.compute_local_smoothing_term_gradient_tikhonov(small, x, y,
                                                                                     copy_if_zero=False)
            kil[y, x], kil_e[y, x] = st.compute_local_smoothing_term_gradient_killing(
                small, x, y, copy_if_zero=False, isomorphic_enforcement_factor=0.1)
            g, e = level_set_term_at_location(live, x, y)
            ls[y, x], ls_e[y, x] = g, e
    out["terms.data_fdm"], out["terms.data_basic"] = data_fdm, data_basic
    out["terms.tikhonov_direct"], out["terms.tikhonov_direct_energy"] = tik, tik_e
    out["terms.killing"], out["terms.killing_energy"] = kil, kil_e
    out["terms.level_set"], out["terms.level_set_energy"] = ls, ls_e

    # 3-D linear resampling prototype (math_utils/resampling.py)
    vol = rng.standard_normal((4, 6, 4))
    out["resampling.vol"] = vol
    out["resampling.up"] = mr.upsample2x_linear(vol)
    out["resampling.down"] = mr.downsample2x_linear(vol)
    np.savez_compressed(os.path.join(HERE, "ref_leaf.npz"), **out)
    print("ref_leaf.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------------------------
def circle_pair(n):
    """2-D analogue of the SURVEY 8(d) 'sphere-pair' generator (closed form)"""
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    c, r, h = n / 2.0, 0.3 * n, 10.0

    def tsdf(sx, sy, ax, ay):
        d = np.sqrt(((xx - (c + sx)) / ax) ** 2 + ((yy - (c + sy)) / ay) ** 2) - r
        return np.clip(d / h, -1.0, 1.0).astype(np.float32)

    return tsdf(0.0, 0.0, 1.0, 1.0), tsdf(1.5, -1.0, 1.05, 0.95)


def hierarchical():
    import tests.test_data.hierarchical_optimizer_test_data as hd
    out = {}
    cases = {"g16": (hd.canonical_field, hd.live_field)}
    c64, l64 = circle_pair(64)
    cases["c64"] = (c64, l64)
    out["c64.canonical"], out["c64.live"] = c64, l64
    k7 = generate_1d_sobolev_kernel(size=7, strength=0.1)
    out["kernel7"] = k7
    captured = []

    def hook(self, level, it, canonical_lvl, resampled_live, warp_field, data_gradient=None,
             inverse_tikhonov_gradient=None):
        captured.append((level, it, warp_field.copy()))

    hov.HierarchicalOptimizer2dVisualizer.generate_per_iteration_visualizations = hook
    for case, (canon, live) in cases.items():
        for tik in (False, True):
            for ker in (False, True):
                for chunk in (4, 8):
                    n_it = 4
                    opt = ho.HierarchicalOptimizer2d(tikhonov_term_enabled=tik, gradient_kernel_enabled=ker,
                                                     maximum_chunk_size=chunk, rate=0.2,
                                                     maximum_iteration_count=n_it,
                                                     maximum_warp_update_threshold=0.0, data_term_amplifier=1.0,
                                                     tikhonov_strength=0.2, kernel=k7 if ker else None)
                    del captured[:]
                    try:
                        warp = opt.optimize(canon, live)
                    except ValueError:
                        # the reference cannot convolve a pyramid level narrower than the kernel
                        # (np.convolve 'same' returns max(M, N) samples) -- no golden for that combination
                        continue
                    tag = "%s.tik%d.ker%d.chunk%d" % (case, tik, ker, chunk)
                    out[tag + ".final_warp"] = warp
                    if case == "g16" or chunk == 8:
                        for level, it, w in captured:
                            out["%s.L%d.it%d.warp" % (tag, level, it)] = w
    # threshold-terminated run (iteration counts are part of the contract)
    opt = ho.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=False,
                                     maximum_chunk_size=8, rate=0.1, maximum_iteration_count=40,
                                     maximum_warp_update_threshold=0.01, tikhonov_strength=0.2)
    del captured[:]
    out["c64.threshold_run.final_warp"] = opt.optimize(c64, l64)
    counts = {}
    for level, it, _ in captured:
        counts[level] = max(counts.get(level, 0), it + 1)
    out["c64.threshold_run.iteration_counts"] = np.array([counts[k] for k in sorted(counts)])
    np.savez_compressed(os.path.join(HERE, "ref_hierarchical.npz"), **out)
    print("ref_hierarchical.npz:", len(out), "arrays")


def config1():
    """BASELINE config 1 at its full length: the reference's 64 x 64 orthographic pair, 100 fixed iterations --
    SobolevFusion (VECTORIZED, Tikhonov + 7-tap Sobolev kernel) and KillingFusion (DIRECT, Killing + level set) -- with
    snapshots on the way (a 3-iteration excerpt is in ref_slavcheva.npz)"""
    out = {}
    sampling.set_focus_coordinates(0, 0)
    live_full, canon_full = tsdf_gen.generate_initial_orthographic_2d_tsdf_fields(field_size=128)
    live64 = live_full[30:94, 30:94].copy()
    canon64 = canon_full[30:94, 30:94].copy()
    k7 = generate_1d_sobolev_kernel(size=7, strength=0.1)
    configs = {
        "sobolev_vec": dict(compute_method=so.ComputeMethod.VECTORIZED, sobolev_smoothing_enabled=True,
                            sobolev_kernel=k7),
        "killing": dict(compute_method=so.ComputeMethod.DIRECT, level_set_term_enabled=True,
                        smoothing_term_method=st.SmoothingTermMethod.KILLING),
    }
    captured = {}

    def hook(self, iteration_number, warp_field, gradient_field, live_field, canonical_field):
        if iteration_number in (9, 24, 49, 99):
            captured[iteration_number] = (warp_field.copy(), live_field.copy())

    sviz.SlavchevaVisualizer.write_all_iteration_visualizations = hook
    tmp = tempfile.mkdtemp()
    for name, kw in configs.items():
        opt = so.SlavchevaOptimizer2d(out_path=tmp, field_size=64, maximum_warp_length_lower_threshold=0.0,
                                      max_iterations=100, min_iterations=100, enable_convergence_status_logging=True, **kw)
        live = live64.copy()
        captured.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            opt.optimize(live, canon64)
        tag = "ortho64.%s.100" % name
        out[tag + ".final_live"] = live
        out[tag + ".final_gradient"] = opt.gradient_field.copy()
        out[tag + ".max_warps"] = np.array(opt.log.max_warps, dtype=np.float64)
        out[tag + ".data_energies"] = np.array(opt.log.data_energies, dtype=np.float64)
        out[tag + ".smoothing_energies"] = np.array(opt.log.smoothing_energies, dtype=np.float64)
        out[tag + ".level_set_energies"] = np.array(opt.log.level_set_energies, dtype=np.float64)
        for it, (w, l) in sorted(captured.items()):
            out["%s.it%d.warp" % (tag, it)] = w
            out["%s.it%d.live" % (tag, it)] = l
        print(tag, "max warp first / last: %.4f / %.4f" % (opt.log.max_warps[0], opt.log.max_warps[-1]))
    np.savez_compressed(os.path.join(HERE, "ref_config1.npz"), **out)
    print("ref_config1.npz:", len(out), "arrays")


def orthographic():
    """BASELINE config 1's INPUT generator: generate_initial_orthographic_2d_tsdf_fields with and without mimic_eta,
    other band widths / default values, a free-standing polyline through generate_sample_orthographic_2d_tsdf_field,
    and which exception a field that cannot hold the surface raises (stored as a code: 1 IndexError, 2 ValueError)"""
    from utils.point2d import Point2d
    out = {}
    for tag, kw in (("size128", dict(field_size=128)),
                    ("size128.eta", dict(field_size=128, mimic_eta=True)),
                    ("size160.band12.default0", dict(field_size=160, narrow_band_width_voxels=12, default_value=0)),
                    ("size128.band30.eta.default_half", dict(field_size=128, narrow_band_width_voxels=30,
                                                             mimic_eta=True, default_value=0.5)),
                    ("size110.band7", dict(field_size=110, narrow_band_width_voxels=7))):
        live, canonical = tsdf_gen.generate_initial_orthographic_2d_tsdf_fields(**kw)
        out[tag + ".live"], out[tag + ".canonical"] = live, canonical
    codes = {IndexError: 1, ValueError: 2}
    for tag, kw in (("size64", dict(field_size=64)), ("size100.eta", dict(field_size=100, mimic_eta=True)),
                    ("size128.band60", dict(field_size=128, narrow_band_width_voxels=60)),
                    ("size128.smoothing", dict(field_size=128, live_smoothing_kernel_size=3))):
        try:
            tsdf_gen.generate_initial_orthographic_2d_tsdf_fields(**kw)
            code = 0
        except NameError:
            code = 3
        except (IndexError, ValueError) as e:
            code = codes[type(e)]
        out[tag + ".raises"] = np.array(code)
    points = np.array([[3.5, 30.25], [10.0, 41.5], [17.25, 33.0], [40.0, 52.125]], dtype=np.float32)
    out["polyline.points"] = points
    for tag, kw in (("polyline.size80", dict(size=80)), ("polyline.size80.cut5", dict(size=80, back_cutoff_voxels=5)),
                    ("polyline.size72.band8.default_m1", dict(size=72, narrow_band_width_voxels=8,
                                                              default_value=-1))):
        out[tag] = tsdf_gen.generate_sample_orthographic_2d_tsdf_field([Point2d(x, y) for x, y in points], **kw)
    # a band that runs past the last row: the rows that exist are written, then IndexError
    field = np.full((40, 48), 0.25, dtype=np.float32)
    try:
        tsdf_gen.add_surface_to_2d_tsdf_field_sample(field, [Point2d(x, y) for x, y in points])
        out["polyline.rows40.raises"] = np.array(0)
    except IndexError:
        out["polyline.rows40.raises"] = np.array(1)
    out["polyline.rows40.partial"] = field
    np.savez_compressed(os.path.join(HERE, "ref_orthographic.npz"), **out)
    print("ref_orthographic.npz:", len(out), "arrays")


def slavcheva():
    out = {}
    sampling.set_focus_coordinates(0, 0)
    live_full, canon_full = tsdf_gen.generate_initial_orthographic_2d_tsdf_fields(field_size=128)
    live32 = live_full[46:78, 40:72].copy()
    canon32 = canon_full[46:78, 40:72].copy()
    live64 = live_full[30:94, 30:94].copy()
    canon64 = canon_full[30:94, 30:94].copy()
    out["ortho32.live"], out["ortho32.canonical"] = live32, canon32
    out["ortho64.live"], out["ortho64.canonical"] = live64, canon64
    k7 = generate_1d_sobolev_kernel(size=7, strength=0.1)
    k3 = generate_1d_sobolev_kernel(size=3, strength=0.1)
    out["kernel7"], out["kernel3"] = k7, k3
    configs = {
        # SobolevFusion: VECTORIZED, Tikhonov + Sobolev
        "sobolev_vec": dict(compute_method=so.ComputeMethod.VECTORIZED, sobolev_smoothing_enabled=True,
                            sobolev_kernel=k7),
        # the reference's own test configuration, DIRECT
        "sobolev_direct": dict(compute_method=so.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                               sobolev_kernel=k3),
        # KillingFusion: DIRECT, Killing + level set, no Sobolev
        "killing": dict(compute_method=so.ComputeMethod.DIRECT, level_set_term_enabled=True,
                        smoothing_term_method=st.SmoothingTermMethod.KILLING),
        "tikhonov_direct": dict(compute_method=so.ComputeMethod.DIRECT),
        "fdm_direct": dict(compute_method=so.ComputeMethod.DIRECT,
                           data_term_method=dt.DataTermMethod.THRESHOLDED_FDM),
    }
    captured = []

    def hook(self, iteration_number, warp_field, gradient_field, live_field, canonical_field):
        captured.append((iteration_number, warp_field.copy(), gradient_field.copy(), live_field.copy()))

    sviz.SlavchevaVisualizer.write_all_iteration_visualizations = hook
    tmp = tempfile.mkdtemp()
    for size_tag, live0, canon, n_it, per_it in (("ortho32", live32, canon32, 4, True),
                                                 ("ortho64", live64, canon64, 3, False)):
        for name, kw in configs.items():
            opt = so.SlavchevaOptimizer2d(out_path=tmp, field_size=live0.shape[0],
                                          maximum_warp_length_lower_threshold=0.0, max_iterations=n_it,
                                          min_iterations=n_it, enable_convergence_status_logging=True, **kw)
            live = live0.copy()
            del captured[:]
            with contextlib.redirect_stdout(io.StringIO()):
                opt.optimize(live, canon)
            tag = "%s.%s" % (size_tag, name)
            out[tag + ".final_live"] = live
            out[tag + ".final_gradient"] = opt.gradient_field.copy()
            out[tag + ".max_warps"] = np.array(opt.log.max_warps, dtype=np.float64)
            out[tag + ".data_energies"] = np.array(opt.log.data_energies, dtype=np.float64)
            out[tag + ".smoothing_energies"] = np.array(opt.log.smoothing_energies, dtype=np.float64)
            out[tag + ".level_set_energies"] = np.array(opt.log.level_set_energies, dtype=np.float64)
            out[tag + ".final_warp"] = captured[-1][1]
            if per_it:
                for it, w, g, l in captured:
                    out["%s.it%d.warp" % (tag, it)] = w
                    out["%s.it%d.gradient" % (tag, it)] = g
                    out["%s.it%d.live" % (tag, it)] = l
    np.savez_compressed(os.path.join(HERE, "ref_slavcheva.npz"), **out)
    print("ref_slavcheva.npz:", len(out), "arrays")


def focus():
    """the "focus neighbourhood" trace of DIRECT runs (slavcheva_optimizer2d.py:319-322,:422-430): the 3 x 3 voxels around
    the module-global focus coordinate -- one inside the band, one in a corner of the field (4 voxels remain)"""
    out = {}
    live_full, canon_full = tsdf_gen.generate_initial_orthographic_2d_tsdf_fields(field_size=128)
    live32 = live_full[46:78, 40:72].copy()
    canon32 = canon_full[46:78, 40:72].copy()
    out["ortho32.live"], out["ortho32.canonical"] = live32, canon32
    k3 = generate_1d_sobolev_kernel(size=3, strength=0.1)
    out["kernel3"] = k3
    configs = {
        "sobolev_direct": dict(compute_method=so.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True, sobolev_kernel=k3),
        "killing": dict(compute_method=so.ComputeMethod.DIRECT, level_set_term_enabled=True,
                        smoothing_term_method=st.SmoothingTermMethod.KILLING),
        "tikhonov_direct": dict(compute_method=so.ComputeMethod.DIRECT),
    }
    sviz.SlavchevaVisualizer.write_all_iteration_visualizations = lambda *a, **k: None
    tmp = tempfile.mkdtemp()
    for fx, fy in ((16, 14), (0, 31)):
        sampling.set_focus_coordinates(fx, fy)
        for name, kw in configs.items():
            opt = so.SlavchevaOptimizer2d(out_path=tmp, field_size=32, maximum_warp_length_lower_threshold=0.0,
                                          max_iterations=6, min_iterations=6, **kw)
            live = live32.copy()
            with contextlib.redirect_stdout(io.StringIO()):
                opt.optimize(live, canon32)
            tag = "ortho32.%s.focus_%d_%d" % (name, fx, fy)
            keys = list(opt.focus_neighborhood_log.keys())
            out[tag + ".keys"] = np.array(keys, dtype=np.int64)
            out[tag + ".warp_magnitudes"] = np.array([opt.focus_neighborhood_log[k].warp_magnitudes for k in keys],
                                                     dtype=np.float32)
            out[tag + ".sdf_values"] = np.array([opt.focus_neighborhood_log[k].sdf_values for k in keys],
                                                dtype=np.float32)
            out[tag + ".canonical_sdf"] = np.array([opt.focus_neighborhood_log[k].canonical_sdf for k in keys],
                                                   dtype=np.float32)
            print(tag, "largest traced update %.5f" % out[tag + ".warp_magnitudes"].max())
    sampling.set_focus_coordinates(0, 0)
    np.savez_compressed(os.path.join(HERE, "ref_focus.npz"), **out)
    print("ref_focus.npz:", len(out), "arrays")


def tsdf():
    """nearest-pixel TSDF generation (tsdf/generation.py:130-207, 356-437) on the closed-form synthetic depth image
    (oracle.synthetic_depth_image; only its checksum is stored, the tests regenerate it)"""
    from calib.camera import Camera, DepthCamera
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import lsf_oracle as O
    out = {}
    K = np.array([[700., 0., 320.], [0., 700., 240.], [0., 0., 1.]], dtype=np.float32)
    cam = DepthCamera(intrinsics=Camera.Intrinsics((480, 640), intrinsic_matrix=K), depth_unit_ratio=0.001)
    cam64 = DepthCamera(intrinsics=Camera.Intrinsics((480, 640), intrinsic_matrix=K.astype(np.float64)),
                        depth_unit_ratio=0.001)
    d0, d1 = O.synthetic_depth_image(), O.synthetic_depth_image(shift_px=2.0, nearer_m=0.008)
    out["depth0.checksum"] = np.array([int(d0.astype(np.int64).sum()), int((d0.astype(np.int64) ** 2).sum())])
    out["depth1.checksum"] = np.array([int(d1.astype(np.int64).sum()), int((d1.astype(np.int64) ** 2).sum())])
    out["intrinsics"] = K
    for tag, d in (("d0", d0), ("d1", d1)):
        for y in (200, 240):
            out["%s.row%d.n32" % (tag, y)] = gen_mod.generate_2d_tsdf_field_from_depth_image_no_interpolation(
                d, cam, y, field_size=32, array_offset=np.array([-16, -16, 234]))
        out["%s.vol16" % tag] = gen_mod.generate_3d_tsdf_field_from_depth_image(
            d, cam, field_size=16, array_offset=np.array([-8, -8, 240]))
    E = np.eye(4, dtype=np.float32)
    E[0, 3], E[2, 3], E[0, 0], E[0, 2], E[2, 0], E[2, 2] = 0.013, -0.02, 0.9998, 0.02, -0.02, 0.9998
    out["extrinsic"] = E
    out["d0.vol12.extrinsic"] = gen_mod.generate_3d_tsdf_field_from_depth_image(
        d0, cam, camera_extrinsic_matrix=E, field_size=12, array_offset=np.array([-6, -6, 244]))
    out["d0.vol12.k64"] = gen_mod.generate_3d_tsdf_field_from_depth_image(
        d0, cam64, field_size=12, array_offset=np.array([-6, -6, 244]))
    out["d0.row240.n32.default0"] = gen_mod.generate_2d_tsdf_field_from_depth_image(
        d0, cam, 240, field_size=32, default_value=0, array_offset=np.array([-16, -16, 234]),
        narrow_band_width_voxels=10)
    # the two bilinear 2-D generators (tsdf/generation.py:18-128): float32 and float64 intrinsics, a row band whose
    # right-most voxels project past the image border (offset +150 voxels in x), default value 0
    for tag, d in (("d0", d0), ("d1", d1)):
        for name, fn in (("bilinear_image", gen_mod.generate_2d_tsdf_field_from_depth_image_bilinear_image_space),
                         ("bilinear_tsdf", gen_mod.generate_2d_tsdf_field_from_depth_image_bilinear_tsdf_space)):
            out["%s.%s.row240.n32" % (tag, name)] = fn(d, cam, 240, field_size=32,
                                                       array_offset=np.array([-16, -16, 234]))
    for name, fn in (("bilinear_image", gen_mod.generate_2d_tsdf_field_from_depth_image_bilinear_image_space),
                     ("bilinear_tsdf", gen_mod.generate_2d_tsdf_field_from_depth_image_bilinear_tsdf_space)):
        out["d0.%s.row200.n32.k64" % name] = fn(d0, cam64, 200, field_size=32, array_offset=np.array([-16, -16, 234]))
        out["d0.%s.row240.n32.border" % name] = fn(d0, cam, 240, field_size=32, default_value=0,
                                                   array_offset=np.array([90, -16, 234]), narrow_band_width_voxels=10)
        out["d0.%s.row240.n32.extrinsic" % name] = fn(d0, cam, 240, camera_extrinsic_matrix=E, field_size=32,
                                                      array_offset=np.array([-16, -16, 234]))
    np.savez_compressed(os.path.join(HERE, "ref_tsdf.npz"), **out)
    print("ref_tsdf.npz:", len(out), "arrays")


def ewa():
    """EWA TSDF generation (tsdf/ewa.py): the reference's own test cases (tests/test_tsdf_ewa.py:40-235: inline depth
    patch, rows of its two synthetic zigzag depth PNGs, expected arrays of tests/test_data/ewa_test_data.py) plus
    outputs of the reference on the closed-form synthetic depth image.  Depth PNG content is stored as a band of rows
    [ROW0, ROW1) (all other rows are 65535 = "no measurement" as far as these cases can see: verified below)."""
    import tsdf.ewa as ewa_mod
    import tests.test_data.ewa_test_data as data
    from calib.camera import Camera, DepthCamera
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import lsf_oracle as O
    out = {}
    K = np.array([[700., 0., 320.], [0., 700., 240.], [0., 0., 1.]], dtype=np.float32)
    cam = DepthCamera(intrinsics=Camera.Intrinsics((480, 640), intrinsic_matrix=K), depth_unit_ratio=0.001)
    ROW0, ROW1 = 150, 260
    out["rows"] = np.array([ROW0, ROW1])

    def load(name):
        d = np.array(Image.open(os.path.join(REF, "tests", "test_data", name)))
        d[d == 0] = np.iinfo(np.uint16).max  # tests/test_tsdf_ewa.py:36-38
        return d

    def banded(d):
        b = np.full_like(d, np.iinfo(np.uint16).max)
        b[ROW0:ROW1] = d[ROW0:ROW1]
        return b

    z1, z2 = load("zigzag1_depth_00064.png"), load("zigzag2_depth_00108.png")
    out["zigzag1.rows"], out["zigzag2.rows"] = z1[ROW0:ROW1].copy(), z2[ROW0:ROW1].copy()
    for name in dir(data):
        v = getattr(data, name)
        if isinstance(v, np.ndarray):
            out["expected." + name] = v
    patch = np.full((3, 640), np.iinfo(np.uint16).max, dtype=np.uint16)
    region = np.array([3233, 3246, 3243, 3256, 3253, 3268, 3263, 3279, 3272, 3289, 3282, 3299, 3291, 3308, 3301, 3317,
                       3310, 3326], dtype=np.uint16)
    patch[:, 399:417] = region  # tests/test_tsdf_ewa.py:41-50 (three identical rows)
    out["patch"] = patch
    off2 = np.array([-256, 0, 0]) + np.array([210, 0, 103])
    cases = {
        "case1.image.patch": lambda d=patch: ewa_mod.generate_tsdf_2d_ewa_image(
            d, cam, 1, field_size=16, array_offset=np.array([94, -256, 804]), voxel_size=0.004),
        "case2.image.zigzag2": lambda d=z2: ewa_mod.generate_tsdf_2d_ewa_image(
            d, cam, 200, field_size=16, array_offset=off2, voxel_size=0.004),
        "case3.voxel.zigzag1": lambda d=z1: ewa_mod.generate_tsdf_2d_ewa_tsdf(
            d, cam, 200, field_size=16, array_offset=np.array([-232, -256, 490]), voxel_size=0.004,
            gaussian_covariance_scale=0.5),
        "case4.inclusive.zigzag1": lambda d=z1: ewa_mod.generate_tsdf_2d_ewa_tsdf_inclusive(
            d, cam, 200, field_size=16, array_offset=np.array([-232, -256, 490]), voxel_size=0.004,
            gaussian_covariance_scale=0.5),
        "case5.image3d.zigzag2": lambda d=z2: ewa_mod.generate_tsdf_3d_ewa_image(
            d, cam, field_shape=np.array([16, 1, 16]), array_offset=np.array([-46, -8, 105]), voxel_size=0.004),
    }
    for key, fn in cases.items():
        out[key] = fn()
    # the stored row band is all these cases can see
    assert np.array_equal(out["case2.image.zigzag2"], cases["case2.image.zigzag2"](banded(z2)))
    assert np.array_equal(out["case3.voxel.zigzag1"], cases["case3.voxel.zigzag1"](banded(z1)))
    assert np.array_equal(out["case4.inclusive.zigzag1"], cases["case4.inclusive.zigzag1"](banded(z1)))
    assert np.array_equal(out["case5.image3d.zigzag2"], cases["case5.image3d.zigzag2"](banded(z2)))
    # synthetic depth: all four generators, incl. a rotated camera and the image border (inclusive variant)
    d0 = O.synthetic_depth_image()
    E = np.eye(4, dtype=np.float32)
    E[0, 3], E[2, 3], E[0, 0], E[0, 2], E[2, 0], E[2, 2] = 0.013, -0.02, 0.9998, 0.02, -0.02, 0.9998
    out["extrinsic"] = E
    out["syn.image2d"] = ewa_mod.generate_tsdf_2d_ewa_image(d0, cam, 240, field_size=20,
                                                           array_offset=np.array([-10, -10, 236]))
    out["syn.voxel2d"] = ewa_mod.generate_tsdf_2d_ewa_tsdf(d0, cam, 240, field_size=20,
                                                          array_offset=np.array([-10, -10, 236]),
                                                          gaussian_covariance_scale=2.0)
    out["syn.inclusive2d.border"] = ewa_mod.generate_tsdf_2d_ewa_tsdf_inclusive(
        d0, cam, 0, field_size=20, array_offset=np.array([100, -10, 236]), gaussian_covariance_scale=2.0)
    out["syn.image3d.extrinsic"] = ewa_mod.generate_tsdf_3d_ewa_image(
        d0, cam, camera_extrinsic_matrix=E, field_shape=np.array([10, 6, 10]), array_offset=np.array([-5, -3, 238]))
    np.savez_compressed(os.path.join(HERE, "ref_ewa.npz"), **out)
    print("ref_ewa.npz:", len(out), "arrays")


if __name__ == "__main__":
    os.chdir(tempfile.mkdtemp())
    if len(sys.argv) > 1 and sys.argv[1] == "tsdf":
        tsdf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ewa":
        ewa()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "config1":
        config1()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "orthographic":
        orthographic()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "focus":
        focus()
        sys.exit(0)
    literals()
    leaf()
    hierarchical()
    slavcheva()
    config1()
    orthographic()
    focus()
    tsdf()
    ewa()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%8d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))
