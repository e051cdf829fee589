"""BASELINE config 1's INPUT generator (SURVEY 8(d) "the reference's own polyline generator re-stated in the build";
reference tsdf/generation.py:238-353): the package's host generator and the oracle's scalar restatement against outputs
of the reference itself (tests/golden/ref_orthographic.npz, generator tests/golden/make_golden.py orthographic) --
bit for bit, errors included -- and the crops every Slavcheva fixture was cut from fall out of it."""
import os

import numpy as np
import pytest

import levelsetfusion_python_amd  # noqa: F401
from levelsetfusion_python_amd.tsdf import generation as G
from oracle import lsf_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
S = np.load(os.path.join(GOLD, "ref_orthographic.npz"))

PAIRS = {
    "size128": dict(field_size=128),
    "size128.eta": dict(field_size=128, mimic_eta=True),
    "size160.band12.default0": dict(field_size=160, narrow_band_width_voxels=12, default_value=0),
    "size128.band30.eta.default_half": dict(field_size=128, narrow_band_width_voxels=30, mimic_eta=True,
                                            default_value=0.5),
    "size110.band7": dict(field_size=110, narrow_band_width_voxels=7),
}
ERRORS = {
    "size64": (dict(field_size=64), IndexError),            # the surface runs to column 108
    "size100.eta": (dict(field_size=100, mimic_eta=True), IndexError),
    "size128.band60": (dict(field_size=128, narrow_band_width_voxels=60), ValueError),   # band wider than the surface's row
    "size128.smoothing": (dict(field_size=128, live_smoothing_kernel_size=3), NameError),  # tsdf/generation.py:345
}
CODES = {IndexError: 1, ValueError: 2, NameError: 3}


@pytest.mark.parametrize("tag", sorted(PAIRS))
def test_package_generator_equals_reference(tag):
    live, canonical = G.generate_initial_orthographic_2d_tsdf_fields(**PAIRS[tag])
    assert live.dtype == np.float32 and canonical.dtype == np.float32
    assert np.array_equal(live, S[tag + ".live"])
    assert np.array_equal(canonical, S[tag + ".canonical"])


@pytest.mark.parametrize("tag", sorted(PAIRS))
def test_oracle_generator_equals_reference(tag):
    live, canonical = O.orthographic_pair(**PAIRS[tag])
    assert np.array_equal(live, S[tag + ".live"])
    assert np.array_equal(canonical, S[tag + ".canonical"])


@pytest.mark.parametrize("tag", sorted(ERRORS))
def test_generator_raises_what_the_reference_raises(tag):
    kw, error = ERRORS[tag]
    assert int(S[tag + ".raises"]) == CODES[error]
    with pytest.raises(error):
        G.generate_initial_orthographic_2d_tsdf_fields(**kw)


def test_free_standing_polylines_and_partial_fill():
    points = [G.Point2d(x, y) for x, y in S["polyline.points"]]
    cases = {"polyline.size80": dict(size=80), "polyline.size80.cut5": dict(size=80, back_cutoff_voxels=5),
             "polyline.size72.band8.default_m1": dict(size=72, narrow_band_width_voxels=8, default_value=-1)}
    for tag, kw in cases.items():
        assert np.array_equal(G.generate_sample_orthographic_2d_tsdf_field(points, **kw), S[tag]), tag
        field = np.full((kw["size"],) * 2, kw.get("default_value", 1), dtype=np.float32)
        O.orthographic_surface_fill(field, [(p.x, p.y) for p in points], kw.get("narrow_band_width_voxels", 20),
                                    kw.get("back_cutoff_voxels", np.inf))
        assert np.array_equal(field, S[tag]), tag
    # plain (x, y) pairs are accepted too
    assert np.array_equal(G.generate_sample_orthographic_2d_tsdf_field(S["polyline.points"], 80), S["polyline.size80"])
    # a band that runs past the last row: the reference writes what exists, then raises
    assert int(S["polyline.rows40.raises"]) == 1
    field = np.full((40, 48), 0.25, dtype=np.float32)
    with pytest.raises(IndexError):
        G.add_surface_to_2d_tsdf_field_sample(field, points)
    assert np.array_equal(field, S["polyline.rows40.partial"])


def test_the_fixture_crops_fall_out_of_the_generator():
    """every Slavcheva fixture pair (ref_slavcheva.npz, ref_config1.npz) is a crop of the 128 x 128 pair: config 1's
    64 x 64 input is [30:94, 30:94], the 32 x 32 pair [46:78, 40:72]"""
    fixtures = np.load(os.path.join(GOLD, "ref_slavcheva.npz"))
    live, canonical = G.generate_initial_orthographic_2d_tsdf_fields(field_size=128)
    assert np.array_equal(live[30:94, 30:94], fixtures["ortho64.live"])
    assert np.array_equal(canonical[30:94, 30:94], fixtures["ortho64.canonical"])
    assert np.array_equal(live[46:78, 40:72], fixtures["ortho32.live"])
    assert np.array_equal(canonical[46:78, 40:72], fixtures["ortho32.canonical"])


@pytest.mark.gpu
def test_config1_from_the_generator_through_the_optimizer():
    """generator -> SlavchevaOptimizer2d (SobolevFusion configuration, 10 iterations) == the reference's own run on its
    own generator's output (ref_config1.npz snapshot after 10 iterations), without touching a stored input"""
    import levelsetfusion_python_amd as lsf
    ref = np.load(os.path.join(GOLD, "ref_config1.npz"))
    kernels = np.load(os.path.join(GOLD, "ref_slavcheva.npz"))
    live_full, canonical_full = G.generate_initial_orthographic_2d_tsdf_fields(field_size=128)
    live = live_full[30:94, 30:94].copy()
    canonical = canonical_full[30:94, 30:94].copy()
    opt = lsf.SlavchevaOptimizer2d(field_size=64, compute_method=lsf.ComputeMethod.VECTORIZED,
                                   sobolev_smoothing_enabled=True, sobolev_kernel=kernels["kernel7"],
                                   maximum_warp_length_lower_threshold=0.0, max_iterations=10, min_iterations=10)
    opt.optimize(live, canonical)
    assert np.abs(live - ref["ortho64.sobolev_vec.100.it9.live"]).max() <= 1e-5   # north-star tolerance, fp32
