"""TSDF generation on the GPU (nearest pixel) against the oracle (bit for bit) and the reference's
own outputs; plus the depth -> TSDF -> optimizer chain staying on the device."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O
from tests.test_oracle_golden import _tsdf_bilinear_cases, _tsdf_cases, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from levelsetfusion_python_amd.tsdf import generation
    return generation


def _call(gen, depth, K, kw):
    cam = gen.DepthCamera(intrinsic_matrix=K, depth_unit_ratio=0.001)
    shape = kw["field_shape"]
    common = dict(camera_extrinsic_matrix=kw.get("camera_extrinsic_matrix"), field_size=shape[0],
                  default_value=kw.get("default_value", 1), array_offset=np.array(kw["array_offset"]),
                  narrow_band_width_voxels=kw.get("narrow_band_width_voxels", 20))
    if len(shape) == 2:
        return gen.generate_2d_tsdf_field_from_depth_image(depth, cam, kw["image_y_coordinate"], **common)
    return gen.generate_3d_tsdf_field_from_depth_image(depth, cam, **common)


def test_tsdf_nearest_matches_oracle_and_reference(gen, ref_tsdf):
    for key, depth, K, kw in _tsdf_cases(ref_tsdf):
        got = _call(gen, depth, K, kw)
        assert got.dtype == np.float32 and got.shape == tuple(kw["field_shape"])
        assert maxdiff(got, O.tsdf_nearest(depth, K, 0.001, **kw)) == 0.0, key
        assert maxdiff(got, ref_tsdf[key]) <= (2.5e-6 if "extrinsic" in key else 0.0), key


def test_tsdf_bilinear_matches_oracle_and_reference(gen, ref_tsdf):
    """the two bilinear 2-D generators (tsdf/generation.py:18-128) through the dispatcher and by name: bit-identical to
    the oracle, and to the reference's own outputs (general extrinsics: within the matvec-order tolerance)"""
    for key, depth, K, tsdf_space, row, kw in _tsdf_bilinear_cases(ref_tsdf):
        cam = gen.DepthCamera(intrinsic_matrix=K, depth_unit_ratio=0.001)
        common = dict(camera_extrinsic_matrix=kw.get("camera_extrinsic_matrix"), field_size=32,
                      default_value=kw.get("default_value", 1), array_offset=np.array(kw["array_offset"]),
                      narrow_band_width_voxels=kw.get("narrow_band_width_voxels", 20))
        method = gen.FilteringMethod.BILINEAR_VOXEL_SPACE if tsdf_space else gen.FilteringMethod.BILINEAR_IMAGE_SPACE
        got = gen.generate_2d_tsdf_field_from_depth_image(depth, cam, row, interpolation_method=method, **common)
        named = (gen.generate_2d_tsdf_field_from_depth_image_bilinear_tsdf_space if tsdf_space
                 else gen.generate_2d_tsdf_field_from_depth_image_bilinear_image_space)(depth, cam, row, **common)
        assert got.dtype == np.float32 and got.shape == (32, 32) and np.array_equal(got, named)
        assert maxdiff(got, O.tsdf_bilinear(depth, K, 0.001, 32, row, tsdf_space, **kw)) == 0.0, key
        assert maxdiff(got, ref_tsdf[key]) <= (2.5e-6 if "extrinsic" in key else 0.0), key
    # a larger slice, both modes, against the oracle
    d = O.synthetic_depth_image(shift_px=1.0)
    K = ref_tsdf["intrinsics"]
    cam = gen.DepthCamera(intrinsic_matrix=K, depth_unit_ratio=0.001)
    for tsdf_space in (False, True):
        method = gen.FilteringMethod.BILINEAR_VOXEL_SPACE if tsdf_space else gen.FilteringMethod.BILINEAR_IMAGE_SPACE
        got = gen.generate_2d_tsdf_field_from_depth_image(d, cam, 100, field_size=200, interpolation_method=method,
                                                          array_offset=np.array([-100, -100, 150]))
        want = O.tsdf_bilinear(d, K, 0.001, 200, 100, tsdf_space, array_offset=(-100, -100, 150))
        assert maxdiff(got, want) == 0.0 and int((np.abs(want) < 1).sum()) > 2000


def test_tsdf_larger_volume_and_edge_cases(gen):
    K = np.array([[700., 0., 320.], [0., 700., 240.], [0., 0., 1.]], dtype=np.float32)
    depth = O.synthetic_depth_image()
    depth[100:140, 300:330] = 0  # holes (no measurement) keep the default value
    kw = dict(field_shape=(64, 64, 64), array_offset=(-32, -32, 218))
    got = _call(gen, depth, K, kw)
    assert maxdiff(got, O.tsdf_nearest(depth, K, 0.001, **kw)) == 0.0
    assert (np.abs(got) < 1).sum() > 10000
    # volume entirely behind the camera / outside the image: everything stays at the default
    behind = _call(gen, depth, K, dict(field_shape=(8, 8, 8), array_offset=(-4, -4, -300)))
    assert np.all(behind == 1.0)
    outside = _call(gen, depth, K, dict(field_shape=(8, 8, 8), array_offset=(4000, -4, 250)))
    assert np.all(outside == 1.0)
    with pytest.raises(ValueError):
        gen.generate_3d_tsdf_field_from_depth_image(depth.astype(np.float32), gen.DepthCamera(intrinsic_matrix=K))
    with pytest.raises(ValueError):
        gen.generate_2d_tsdf_field_from_depth_image(depth, gen.DepthCamera(intrinsic_matrix=K), 240,
                                                    interpolation_method="NONE")


def test_depth_to_tsdf_to_optimizer_on_device(gen):
    """two synthetic depth frames -> 64^3 TSDF pair -> hierarchical optimizer, nothing leaves the GPU in between;
    result equals the oracle chain bit for bit"""
    import levelsetfusion_python_amd as lsf
    K = np.array([[700., 0., 320.], [0., 700., 240.], [0., 0., 1.]], dtype=np.float32)
    cam = gen.DepthCamera(intrinsic_matrix=K, depth_unit_ratio=0.001)
    d0, d1 = O.synthetic_depth_image(), O.synthetic_depth_image(shift_px=2.0, nearer_m=0.008)
    off = np.array([-32, -32, 218])
    canonical = gen.generate_3d_tsdf_field_from_depth_image(d0, cam, field_size=64, array_offset=off, as_tensor=True)
    live = gen.generate_3d_tsdf_field_from_depth_image(d1, cam, field_size=64, array_offset=off, as_tensor=True)
    assert canonical.is_cuda and live.is_cuda
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=4, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    warp = lsf.HierarchicalOptimizer3d(**kw).optimize(canonical, live)
    assert warp.is_cuda and warp.shape == (64, 64, 64, 3)
    c_ref = O.tsdf_nearest(d0, K, 0.001, (64, 64, 64), array_offset=tuple(off))
    l_ref = O.tsdf_nearest(d1, K, 0.001, (64, 64, 64), array_offset=tuple(off))
    assert maxdiff(warp.cpu().numpy(), O.HierarchicalOracle(**kw).optimize(c_ref, l_ref)) == 0.0
    assert float(warp.abs().max()) > 1e-3


def test_synthetic_depth_pair_for_the_bench(gen):
    """synthetic.depth_pair (bench.py --data depth): two device-resident TSDF volumes, equal to the oracle's generator on
    the same frames, with a band that the optimizer can work on"""
    from levelsetfusion_python_amd import synthetic
    n = 64
    canonical, live = synthetic.depth_pair(n)
    assert canonical.is_cuda and canonical.shape == (n, n, n) and canonical.dtype == torch.float32
    K = np.array([[700., 0., 320.], [0., 700., 240.], [0., 0., 1.]], dtype=np.float32)
    off = (-n // 2, -n // 2, 250 - n // 2)
    for field, kw in ((canonical, dict()), (live, dict(shift_px=2.0, nearer_m=0.008))):
        want = O.tsdf_nearest(O.synthetic_depth_image(**kw), K, 0.001, (n, n, n), array_offset=off)
        assert maxdiff(field.cpu().numpy(), want) == 0.0
    assert int(((canonical.abs() < 1) | (live.abs() < 1)).sum()) > 20000 and not torch.equal(canonical, live)


def test_tsdf_ewa_matches_oracle_and_reference(gen, ref_ewa):
    """the four EWA generators against the oracle (float64 sums; exp / 2x2 inverse may differ in the last place from
    glibc / LAPACK: tolerance 2e-6) and against the reference's own known answers (its tolerance: atol=2e-5)"""
    from levelsetfusion_python_amd.tsdf import ewa
    from tests.test_oracle_golden import _ewa_cases
    K, cases = _ewa_cases(ref_ewa)
    cam = gen.DepthCamera(intrinsic_matrix=K, depth_unit_ratio=0.001)
    fns = {O.EWA_IMAGE: ewa.generate_tsdf_2d_ewa_image, O.EWA_VOXEL: ewa.generate_tsdf_2d_ewa_tsdf,
           O.EWA_VOXEL_INCLUSIVE: ewa.generate_tsdf_2d_ewa_tsdf_inclusive}
    for key, depth, kw, expected in cases:
        shape = kw["field_shape"]
        common = dict(camera_extrinsic_matrix=kw.get("camera_extrinsic_matrix"), array_offset=np.array(kw["array_offset"]),
                      gaussian_covariance_scale=kw.get("gaussian_covariance_scale", 1.0))
        if len(shape) == 2:
            got = fns[kw["method"]](depth, cam, kw["image_y_coordinate"], field_size=shape[0], **common)
        else:
            got = ewa.generate_tsdf_3d_ewa_image(depth, cam, field_shape=np.array(shape), **common)
        assert got.shape == tuple(shape) and got.dtype == np.float32
        assert maxdiff(got, O.tsdf_ewa(depth, K, 0.001, **kw)) <= 2e-6, key
        assert maxdiff(got, ref_ewa[key]) <= 4e-6, key
        if expected is not None:
            assert maxdiff(got, ref_ewa[expected]) <= 2e-5, key
    # dispatcher (tsdf/generation.py:219-235) with smoothing_coefficient
    key, depth, kw, _ = cases[2]
    got = gen.generate_2d_tsdf_field_from_depth_image(depth, cam, 200, field_size=16,
                                                      array_offset=np.array(kw["array_offset"]),
                                                      interpolation_method=gen.FilteringMethod.EWA_VOXEL_SPACE,
                                                      smoothing_coefficient=0.5)
    assert maxdiff(got, ref_ewa[key]) <= 4e-6
