"""EWA (elliptical weighted average) TSDF generation on the GPU -- reference: tsdf/ewa.py:59-184 (3-D, image space),
:230-353 (2-D image space), :358-481 (2-D voxel space), :485-624 (2-D voxel space, inclusive),
math_utils/elliptical_gaussians.py.  Same function names and keyword arguments; `as_tensor=True` keeps the result on
the GPU.  Kernel: csrc/lsf_tsdf.hip::tsdf_ewa_kernel."""
import numpy as np

from .generation import FilteringMethod, _generate

near_clipping_distance = 0.05


def _run(method, depth_image, camera, field_shape, image_y_coordinate, camera_extrinsic_matrix, default_value,
         voxel_size, array_offset, narrow_band_width_voxels, gaussian_covariance_scale, as_tensor):
    f = _generate(depth_image, camera, tuple(int(s) for s in field_shape), image_y_coordinate,
                  camera_extrinsic_matrix, default_value, voxel_size, array_offset, narrow_band_width_voxels, method,
                  gaussian_covariance_scale)
    return f if as_tensor else f.cpu().numpy()


def generate_tsdf_2d_ewa_image(depth_image, camera, image_y_coordinate, camera_extrinsic_matrix=None, field_size=128,
                               default_value=1, voxel_size=0.004, array_offset=np.array([-64, -64, 64]),
                               narrow_band_width_voxels=20, back_cutoff_voxels=np.inf, gaussian_covariance_scale=1.0,
                               as_tensor=False):
    """EWA average of DEPTH over the projected ellipse, then TSDF of the averaged depth (tsdf/ewa.py:230-353)"""
    return _run(FilteringMethod.EWA_IMAGE_SPACE, depth_image, camera, (field_size, field_size), image_y_coordinate,
                camera_extrinsic_matrix, default_value, voxel_size, array_offset, narrow_band_width_voxels,
                gaussian_covariance_scale, as_tensor)


def generate_tsdf_2d_ewa_tsdf(depth_image, camera, image_y_coordinate, camera_extrinsic_matrix=None, field_size=128,
                              default_value=1, voxel_size=0.004, array_offset=np.array([-64, -64, 64]),
                              narrow_band_width_voxels=20, back_cutoff_voxels=np.inf, gaussian_covariance_scale=1.0,
                              as_tensor=False):
    """EWA average of per-pixel TSDF values (tsdf/ewa.py:358-481)"""
    return _run(FilteringMethod.EWA_VOXEL_SPACE, depth_image, camera, (field_size, field_size), image_y_coordinate,
                camera_extrinsic_matrix, default_value, voxel_size, array_offset, narrow_band_width_voxels,
                gaussian_covariance_scale, as_tensor)


def generate_tsdf_2d_ewa_tsdf_inclusive(depth_image, camera, image_y_coordinate, camera_extrinsic_matrix=None,
                                        field_size=128, default_value=1, voxel_size=0.004,
                                        array_offset=np.array([-64, -64, 64]), narrow_band_width_voxels=20,
                                        back_cutoff_voxels=np.inf, gaussian_covariance_scale=1.0, as_tensor=False):
    """as generate_tsdf_2d_ewa_tsdf, pixels outside the image count as TSDF 1 (tsdf/ewa.py:485-624)"""
    return _run(FilteringMethod.EWA_VOXEL_SPACE_INCLUSIVE, depth_image, camera, (field_size, field_size),
                image_y_coordinate, camera_extrinsic_matrix, default_value, voxel_size, array_offset,
                narrow_band_width_voxels, gaussian_covariance_scale, as_tensor)


def generate_tsdf_3d_ewa_image(depth_image, camera, camera_extrinsic_matrix=None,
                               field_shape=np.array([128, 128, 128]), default_value=1, voxel_size=0.004,
                               array_offset=np.array([-64, -64, 64]), narrow_band_width_voxels=20,
                               back_cutoff_voxels=np.inf, gaussian_covariance_scale=1.0, as_tensor=False):
    """3-D image-space EWA (tsdf/ewa.py:59-184).  field[a][b][c]: world x on array axis 0, depth on array axis 2 -- the
    reference's axis flip.  (The reference indexes axis 0 with range(field_shape[2]): it only works for
    field_shape[0] == field_shape[2]; any shape works here.)"""
    return _run(FilteringMethod.EWA_IMAGE_SPACE, depth_image, camera, field_shape, None, camera_extrinsic_matrix,
                default_value, voxel_size, array_offset, narrow_band_width_voxels, gaussian_covariance_scale, as_tensor)


generate_tsdf_2d_ewa_functions = {
    FilteringMethod.EWA_IMAGE_SPACE: generate_tsdf_2d_ewa_image,
    FilteringMethod.EWA_VOXEL_SPACE: generate_tsdf_2d_ewa_tsdf,
    FilteringMethod.EWA_VOXEL_SPACE_INCLUSIVE: generate_tsdf_2d_ewa_tsdf_inclusive,
}
