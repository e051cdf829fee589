"""The "focus neighbourhood" trace (slavcheva_optimizer2d.py:48-55, :319-322, :422-430): per voxel of the 3 x 3 (3 x 3 x 3)
block around a focus coordinate, every iteration's live value BEFORE the re-warp and the length of the update BEFORE the
snap of warp_field_advanced.

* CPU: the oracle's restatement against traces of the REFERENCE ITSELF (tests/golden/ref_focus.npz, made by
  tests/golden/make_golden.py focus): a focus inside the band and one in a corner of the field (4 voxels remain; with the
  Killing term the reference's wrap-around there makes updates of 8 voxels).
* GPU: the optimizers' opt-in `focus_coordinates=` against the same fixtures (1e-5, the north-star tolerance) and, in 2-D
  and 3-D, against the oracle bit for bit -- and the call's other results are those of an untraced call."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O
from conftest import load_golden

ATOL = 1e-5
CONFIGS = {
    "sobolev_direct": dict(compute_method=O.DIRECT, sobolev_smoothing_enabled=True),
    "killing": dict(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING),
    "tikhonov_direct": dict(compute_method=O.DIRECT),
}
FOCI = [(16, 14), (0, 31)]


@pytest.fixture(scope="module")
def ref_focus():
    return load_golden("ref_focus.npz")


def _oracle_trace(F, name, focus, iterations=6):
    kw = dict(CONFIGS[name])
    if kw.get("sobolev_smoothing_enabled"):
        kw["sobolev_kernel"] = F["kernel3"]
    o = O.SlavchevaOracle(maximum_warp_length_lower_threshold=0.0, max_iterations=iterations, min_iterations=iterations,
                          **kw)
    keys = [tuple(int(c) for c in k) for k in F["ortho32.%s.focus_%d_%d.keys" % ((name,) + focus)]]
    o.focus_voxels = [k[::-1] for k in keys]
    o.optimize(F["ortho32.live"].copy(), F["ortho32.canonical"])
    return keys, o


@pytest.mark.parametrize("focus", FOCI)
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_trace_is_the_references(ref_focus, name, focus):
    F = ref_focus
    tag = "ortho32.%s.focus_%d_%d" % ((name,) + focus)
    keys, o = _oracle_trace(F, name, focus)
    assert len(keys) == (9 if focus == (16, 14) else 4)
    for v, key in enumerate(keys):
        entry = o.focus_log[key[::-1]]
        assert entry["canonical_sdf"] == F[tag + ".canonical_sdf"][v]
        assert np.abs(np.float64(entry["warp_magnitudes"]) - F[tag + ".warp_magnitudes"][v]).max() <= 2.5e-6
        assert np.abs(np.float64(entry["sdf_values"]) - F[tag + ".sdf_values"][v]).max() <= 2.5e-6
    assert F[tag + ".warp_magnitudes"].max() > 0.04  # the traced voxels moved


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


def _product_kwargs(lsf, name, kernel):
    kw = dict(compute_method=lsf.ComputeMethod.DIRECT)
    if name == "sobolev_direct":
        kw.update(sobolev_smoothing_enabled=True, sobolev_kernel=kernel)
    if name == "killing":
        kw.update(level_set_term_enabled=True, smoothing_term_method=lsf.SmoothingTermMethod.KILLING)
    return kw


@pytest.mark.gpu
@pytest.mark.parametrize("focus", FOCI)
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_traced_call_2d_against_the_reference_and_the_oracle(lsf, ref_focus, tmp_path, name, focus):
    F = ref_focus
    tag = "ortho32.%s.focus_%d_%d" % ((name,) + focus)
    keys, o = _oracle_trace(F, name, focus)
    kw = _product_kwargs(lsf, name, F["kernel3"])
    fixed = dict(out_path=str(tmp_path), field_size=32, maximum_warp_length_lower_threshold=0.0, max_iterations=6,
                 min_iterations=6)
    opt = lsf.SlavchevaOptimizer2d(focus_coordinates=focus, **fixed, **kw)
    live = F["ortho32.live"].copy()
    opt.optimize(live, F["ortho32.canonical"])
    assert list(opt.focus_neighborhood_log.keys()) == keys  # the reference's keys, in its order (x fastest)
    for v, key in enumerate(keys):
        mine, theirs = opt.focus_neighborhood_log[key], o.focus_log[key[::-1]]
        assert mine.canonical_sdf == theirs["canonical_sdf"]
        assert np.array_equal(np.float32(mine.warp_magnitudes), np.float32(theirs["warp_magnitudes"])), key
        assert np.array_equal(np.float32(mine.sdf_values), np.float32(theirs["sdf_values"])), key
        assert np.abs(np.float64(mine.warp_magnitudes) - F[tag + ".warp_magnitudes"][v]).max() <= ATOL
        assert np.abs(np.float64(mine.sdf_values) - F[tag + ".sdf_values"][v]).max() <= ATOL
    # the trace only looks: everything else is what an untraced call gives
    plain = lsf.SlavchevaOptimizer2d(**fixed, **kw)
    live_plain = F["ortho32.live"].copy()
    plain.optimize(live_plain, F["ortho32.canonical"])
    assert plain.focus_neighborhood_log is None
    assert np.array_equal(live, live_plain) and np.array_equal(opt.warp_field, plain.warp_field)
    assert opt.log.max_warps == plain.log.max_warps and opt.log.data_energies == plain.log.data_energies


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["killing", "sobolev_direct"])
def test_traced_call_3d_against_the_oracle(lsf, name):
    """27 voxels around a focus on the sphere pair's surface, device tensors in, a threshold-terminated loop"""
    n, focus = 24, (12, 12, 5)
    canonical, live0 = O.sphere_pair(n, 3)
    kernel = O.generate_1d_sobolev_kernel(3, 0.1)
    okw = dict(CONFIGS[name])
    if okw.get("sobolev_smoothing_enabled"):
        okw["sobolev_kernel"] = kernel
    loop = dict(maximum_warp_length_lower_threshold=0.03, max_iterations=12, min_iterations=2)
    o = O.SlavchevaOracle(**loop, **okw)
    opt = lsf.SlavchevaOptimizer3d(field_size=n, focus_coordinates=focus, **loop, **_product_kwargs(lsf, name, kernel))
    keys = opt._focus_neighbourhood((n, n, n))
    assert len(keys) == 27 and keys[0] == (11, 11, 4) and keys[1] == (12, 11, 4) and keys[-1] == (13, 13, 6)
    o.focus_voxels = [k[::-1] for k in keys]
    o.optimize(live0.copy(), canonical)
    live = torch.from_numpy(live0).cuda()
    opt.optimize(live, torch.from_numpy(canonical).cuda())
    iterations = len(o.log["max_warps"])
    assert 2 <= iterations and np.array_equal(np.float32(opt.log.max_warps), np.float32(o.log["max_warps"]))
    moved = 0.0
    for key in keys:
        mine, theirs = opt.focus_neighborhood_log[key], o.focus_log[key[::-1]]
        assert len(mine.warp_magnitudes) == iterations
        assert mine.canonical_sdf == theirs["canonical_sdf"]
        assert np.array_equal(np.float32(mine.warp_magnitudes), np.float32(theirs["warp_magnitudes"])), key
        assert np.array_equal(np.float32(mine.sdf_values), np.float32(theirs["sdf_values"])), key
        moved = max(moved, float(np.max(mine.warp_magnitudes)))
    assert moved > 0.01


@pytest.mark.gpu
def test_focus_coordinates_are_checked(lsf, tmp_path):
    opt = lsf.SlavchevaOptimizer2d(out_path=str(tmp_path), field_size=16, focus_coordinates=(1, 2, 3))
    field = np.ones((16, 16), dtype=np.float32)
    with pytest.raises(ValueError):
        opt.optimize(field.copy(), field)
