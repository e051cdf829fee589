"""TSDF generation from depth images on the GPU -- the input stage of the optimizers
(reference: tsdf/generation.py:130-235, 356-437; camera model calib/camera.py:69-311 reduced to what is consumed:
a 3x3 intrinsic matrix and the depth unit ratio).  Nearest-pixel lookup (FilteringMethod.NONE) and the two bilinear 2-D
variants (tsdf/generation.py:18-128) here, the EWA filters in tsdf/ewa.py of this package."""
import ctypes
from enum import Enum

import numpy as np
import torch

from .. import _lib, device as dev
from ..engine import as_device_field  # noqa: F401  (require_gpu side effect lives in device.require_gpu)
# the hard-coded piecewise-linear pairs of tsdf/generation.py:238-353 (host numpy; BASELINE config 1's input)
from .orthographic import (Point2d, add_surface_to_2d_tsdf_field_sample,  # noqa: F401
                           generate_initial_orthographic_2d_tsdf_fields, generate_sample_orthographic_2d_tsdf_field)


class FilteringMethod(Enum):
    """mirrors cpp.tsdf.FilteringMethod (tsdf/generation.py:210-217, tsdf/ewa.py:627-631)"""
    NONE = 0
    BILINEAR_IMAGE_SPACE = 1
    BILINEAR_VOXEL_SPACE = 2
    EWA_IMAGE_SPACE = 3
    EWA_VOXEL_SPACE = 4
    EWA_VOXEL_SPACE_INCLUSIVE = 5


class DepthCamera:
    """what the generators read from calib.camera.DepthCamera: .intrinsics.intrinsic_matrix, .depth_unit_ratio"""

    class Intrinsics:
        def __init__(self, resolution=None, intrinsic_matrix=None):
            self.resolution = resolution
            self.intrinsic_matrix = np.eye(3) if intrinsic_matrix is None else intrinsic_matrix

    def __init__(self, resolution=None, intrinsics=None, intrinsic_matrix=None, depth_unit_ratio=0.001):
        self.resolution = resolution
        self.intrinsics = intrinsics if intrinsics is not None else DepthCamera.Intrinsics(resolution,
                                                                                           intrinsic_matrix)
        self.depth_unit_ratio = depth_unit_ratio


def _generate(depth_image, camera, field_shape, image_y_coordinate, camera_extrinsic_matrix, default_value,
              voxel_size, array_offset, narrow_band_width_voxels, ewa_method=None, gaussian_covariance_scale=1.0,
              bilinear_method=None):
    dev.require_gpu()
    P = np.asarray(camera.intrinsics.intrinsic_matrix)
    if isinstance(depth_image, torch.Tensor):
        depth = depth_image.to("cuda")
        if depth.dtype != torch.uint16:
            raise ValueError("depth image tensor must be uint16")
    else:
        d = np.asarray(depth_image)
        if d.dtype != np.uint16:
            raise ValueError("depth image must be uint16 (raw sensor units), got %s" % d.dtype)
        depth = torch.from_numpy(np.ascontiguousarray(d).view(np.int16)).to("cuda").view(torch.uint16)
    if depth.dim() != 2:
        raise ValueError("depth image must be 2-D")
    depth = depth.contiguous()
    E = np.eye(4, dtype=np.float32) if camera_extrinsic_matrix is None else \
        np.asarray(camera_extrinsic_matrix, dtype=np.float32)
    params = _lib.TsdfParams()
    params.intrinsics[:] = [float(P[0, 0]), float(P[1, 1]), float(P[0, 2]), float(P[1, 2])]
    params.depth_unit_ratio = float(camera.depth_unit_ratio)
    params.voxel_size = float(voxel_size)
    params.narrow_band_half_width = narrow_band_width_voxels / 2 * voxel_size
    params.extrinsic[:] = [float(v) for v in E.reshape(-1)]
    params.array_offset[:] = [int(v) for v in array_offset]
    params.image_height, params.image_width = int(depth.shape[0]), int(depth.shape[1])
    params.image_y_coordinate = int(image_y_coordinate) if image_y_coordinate is not None else 0
    params.default_value = float(default_value)
    params.intrinsics_are_f32 = int(P.dtype == np.float32)
    grid = dev.make_grid(field_shape)
    field = torch.empty(tuple(field_shape), dtype=torch.float32, device="cuda")
    if bilinear_method is not None:
        _lib.check(_lib.lib.lsf_tsdf_generate_bilinear(ctypes.c_void_p(depth.data_ptr()),
                                                       ctypes.c_void_p(field.data_ptr()), ctypes.byref(grid),
                                                       ctypes.byref(params), int(bilinear_method.value),
                                                       dev.stream_ptr()), "lsf_tsdf_generate_bilinear")
        return field
    if ewa_method is None:
        _lib.check(_lib.lib.lsf_tsdf_generate_nearest(ctypes.c_void_p(depth.data_ptr()),
                                                      ctypes.c_void_p(field.data_ptr()), ctypes.byref(grid),
                                                      ctypes.byref(params), dev.stream_ptr()),
                   "lsf_tsdf_generate_nearest")
        return field
    ewa = _lib.EwaParams()
    rotation = E[0:3, 0:3]
    # float32 rotation, float64 sphere covariance: tsdf/ewa.py:103-106
    cov = rotation.dot(np.eye(3) * (gaussian_covariance_scale * voxel_size)).dot(rotation.T)
    ewa.covariance_camera_space[:] = [float(v) for v in np.asarray(cov, dtype=np.float64).reshape(-1)]
    ewa.squared_radius_threshold = 4.0 * gaussian_covariance_scale * voxel_size
    ewa.intrinsic_matrix[:] = [float(v) for v in np.asarray(P, dtype=np.float32).reshape(-1)]
    ewa.method = int(ewa_method.value)
    _lib.check(_lib.lib.lsf_tsdf_generate_ewa(ctypes.c_void_p(depth.data_ptr()), ctypes.c_void_p(field.data_ptr()),
                                              ctypes.byref(grid), ctypes.byref(params), ctypes.byref(ewa),
                                              dev.stream_ptr()), "lsf_tsdf_generate_ewa")
    return field


def generate_2d_tsdf_field_from_depth_image_no_interpolation(depth_image, camera, image_y_coordinate,
                                                             camera_extrinsic_matrix=None, field_size=128,
                                                             default_value=1, voxel_size=0.004,
                                                             array_offset=np.array([-64, -64, 64]),
                                                             narrow_band_width_voxels=20, back_cutoff_voxels=np.inf,
                                                             as_tensor=False):
    """(field_size, field_size) float32 slice: x from the x index, depth axis from the y index, depth row
    image_y_coordinate (tsdf/generation.py:130-207)"""
    f = _generate(depth_image, camera, (field_size, field_size), image_y_coordinate, camera_extrinsic_matrix,
                  default_value, voxel_size, array_offset, narrow_band_width_voxels)
    return f if as_tensor else f.cpu().numpy()


def generate_2d_tsdf_field_from_depth_image_bilinear_image_space(depth_image, camera, image_y_coordinate,
                                                                 camera_extrinsic_matrix=None, field_size=128,
                                                                 default_value=1, voxel_size=0.004,
                                                                 array_offset=np.array([-64, -64, 64]),
                                                                 narrow_band_width_voxels=20,
                                                                 back_cutoff_voxels=np.inf, as_tensor=False):
    """depth blended between the two pixels around the voxel's projection, then one TSDF value
    (tsdf/generation.py:78-128)"""
    f = _generate(depth_image, camera, (field_size, field_size), image_y_coordinate, camera_extrinsic_matrix,
                  default_value, voxel_size, array_offset, narrow_band_width_voxels,
                  bilinear_method=FilteringMethod.BILINEAR_IMAGE_SPACE)
    return f if as_tensor else f.cpu().numpy()


def generate_2d_tsdf_field_from_depth_image_bilinear_tsdf_space(depth_image, camera, image_y_coordinate,
                                                                camera_extrinsic_matrix=None, field_size=128,
                                                                default_value=1, voxel_size=0.004,
                                                                array_offset=np.array([-64, -64, 64]),
                                                                narrow_band_width_voxels=20,
                                                                back_cutoff_voxels=np.inf, as_tensor=False):
    """a TSDF value per pixel around the voxel's projection, then blended (tsdf/generation.py:18-75)"""
    f = _generate(depth_image, camera, (field_size, field_size), image_y_coordinate, camera_extrinsic_matrix,
                  default_value, voxel_size, array_offset, narrow_band_width_voxels,
                  bilinear_method=FilteringMethod.BILINEAR_VOXEL_SPACE)
    return f if as_tensor else f.cpu().numpy()


def generate_2d_tsdf_field_from_depth_image(depth_image, camera, image_y_coordinate, camera_extrinsic_matrix=None,
                                            field_size=128, default_value=1, voxel_size=0.004,
                                            array_offset=np.array([-64, -64, 64]), narrow_band_width_voxels=20,
                                            back_cutoff_voxels=np.inf, interpolation_method=FilteringMethod.NONE,
                                            smoothing_coefficient=1.0, as_tensor=False):
    """dispatcher of tsdf/generation.py:219-235"""
    if not isinstance(interpolation_method, FilteringMethod):
        raise ValueError("Unrecognized GenerationMethod enum value: " + str(interpolation_method))
    if interpolation_method in (FilteringMethod.EWA_IMAGE_SPACE, FilteringMethod.EWA_VOXEL_SPACE,
                                FilteringMethod.EWA_VOXEL_SPACE_INCLUSIVE):
        f = _generate(depth_image, camera, (field_size, field_size), image_y_coordinate, camera_extrinsic_matrix,
                      default_value, voxel_size, array_offset, narrow_band_width_voxels, interpolation_method,
                      smoothing_coefficient)
        return f if as_tensor else f.cpu().numpy()
    if interpolation_method in (FilteringMethod.BILINEAR_IMAGE_SPACE, FilteringMethod.BILINEAR_VOXEL_SPACE):
        f = _generate(depth_image, camera, (field_size, field_size), image_y_coordinate, camera_extrinsic_matrix,
                      default_value, voxel_size, array_offset, narrow_band_width_voxels,
                      bilinear_method=interpolation_method)
        return f if as_tensor else f.cpu().numpy()
    return generate_2d_tsdf_field_from_depth_image_no_interpolation(
        depth_image, camera, image_y_coordinate, camera_extrinsic_matrix, field_size, default_value, voxel_size,
        array_offset, narrow_band_width_voxels, back_cutoff_voxels, as_tensor)


def generate_3d_tsdf_field_from_depth_image(depth_image, camera, camera_extrinsic_matrix=None, field_size=128,
                                            default_value=1, voxel_size=0.004, array_offset=np.array([-64, -64, 64]),
                                            narrow_band_width_voxels=20, back_cutoff_voxels=np.inf, as_tensor=False):
    """(field_size,)*3 float32 volume [z][y][x] (tsdf/generation.py:356-437); as_tensor=True keeps it on the GPU,
    ready for HierarchicalOptimizer3d / SlavchevaOptimizer3d without a host round trip"""
    f = _generate(depth_image, camera, (field_size,) * 3, None, camera_extrinsic_matrix, default_value, voxel_size,
                  array_offset, narrow_band_width_voxels)
    return f if as_tensor else f.cpu().numpy()
