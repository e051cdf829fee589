"""Torch-facing wrappers around the C ABI (device.py re-exports them): layouts, field warping (a1-a3), gradients and pyramids
(a4-a6), the separable filter (a9 / a10)."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .device_core import _gate_ref, _ptr, make_grid, n_voxels, stream_ptr

# ---------------------------------------------------------------------------------------------- layout
def deinterleave(interleaved, channels):
    n = interleaved.numel() // channels
    out = torch.empty((channels,) + tuple(interleaved.shape[:-1]), dtype=torch.float32, device=interleaved.device)
    check(lib.lsf_deinterleave(_ptr(interleaved, n * channels, "interleaved"), _ptr(out, n * channels, "planar"),
                               n, channels, stream_ptr()), "lsf_deinterleave")
    return out


def interleave(planar):
    channels = planar.shape[0]
    n = planar.numel() // channels
    out = torch.empty(tuple(planar.shape[1:]) + (channels,), dtype=torch.float32, device=planar.device)
    check(lib.lsf_interleave(_ptr(planar, n * channels, "planar"), _ptr(out, n * channels, "interleaved"), n,
                             channels, stream_ptr()), "lsf_interleave")
    return out


def halo_copy(scalar, planar, msg_lo, msg_hi, halo, z_lo, z_hi, unpack):
    """pack (unpack=False) / unpack the halo messages of a scalar field [z,y,x] and a planar vector field [c,z,y,x]"""
    ref = scalar if scalar is not None else planar[0]
    grid = make_grid(ref.shape)
    planes = 0 if planar is None else planar.shape[0]
    n = n_voxels(grid)
    per_msg = (1 + planes) * halo * grid.ny * grid.nx
    check(lib.lsf_halo_copy(_ptr(scalar, n, "scalar", allow_none=True),
                            _ptr(planar, n * planes, "planar", allow_none=True) if planar is not None
                            else ctypes.c_void_p(0),
                            _ptr(msg_lo, per_msg, "msg_lo", allow_none=True),
                            _ptr(msg_hi, per_msg, "msg_hi", allow_none=True), ctypes.byref(grid), planes, int(halo),
                            int(z_lo), int(z_hi), int(bool(unpack)), stream_ptr()), "lsf_halo_copy")


# ---------------------------------------------------------------------------------------------- a1-a3
def warp_field(field, warp_planar, oob_value, grid=None, out=None):
    grid = grid or make_grid(field.shape)
    n = n_voxels(grid)
    out = torch.empty_like(field) if out is None else out
    check(lib.lsf_warp_field(_ptr(field, n, "field"), _ptr(warp_planar, n * grid.dims, "warp"),
                             _ptr(out, n, "out"), ctypes.byref(grid), float(oob_value), stream_ptr()),
          "lsf_warp_field")
    return out


def warp_field_advanced(canonical, live, warp_planar, gradient_planar, flags, grid=None, out=None):
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    out = torch.empty_like(live) if out is None else out
    check(lib.lsf_warp_field_advanced(_ptr(canonical, n, "canonical"), _ptr(live, n, "live"),
                                      _ptr(warp_planar, n * grid.dims, "warp"),
                                      _ptr(gradient_planar, n * grid.dims, "gradient", allow_none=True),
                                      _ptr(out, n, "new_live"), ctypes.byref(grid), int(flags), stream_ptr()),
          "lsf_warp_field_advanced")
    return out


# ---------------------------------------------------------------------------------------------- a4-a6
def pack_live_gradient(live, grid=None):
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    out = torch.empty(tuple(live.shape) + (4,), dtype=torch.float32, device=live.device)
    check(lib.lsf_pack_live_gradient(_ptr(live, n, "live"), _ptr(out, 4 * n, "packed"), ctypes.byref(grid),
                                     stream_ptr()), "lsf_pack_live_gradient")
    return out


def restrict_mean(fine, channels):
    """fine: [z,]y,x (channels == 1) or [z,]y,x,4"""
    spatial = tuple(fine.shape) if channels == 1 else tuple(fine.shape[:-1])
    grid = make_grid(spatial)
    coarse_spatial = tuple(s // 2 for s in spatial)
    out = torch.empty(coarse_spatial + (() if channels == 1 else (channels,)), dtype=torch.float32,
                      device=fine.device)
    check(lib.lsf_restrict_mean(_ptr(fine, n_voxels(grid) * channels, "fine"),
                                _ptr(out, out.numel(), "coarse"), ctypes.byref(grid), channels, stream_ptr()),
          "lsf_restrict_mean")
    return out


def prolong_repeat(coarse_planar):
    dims = coarse_planar.shape[0]
    fine_spatial = tuple(2 * s for s in coarse_planar.shape[1:])
    grid = make_grid(fine_spatial)
    out = torch.empty((dims,) + fine_spatial, dtype=torch.float32, device=coarse_planar.device)
    check(lib.lsf_prolong_repeat(_ptr(coarse_planar, coarse_planar.numel(), "coarse"),
                                 _ptr(out, n_voxels(grid) * dims, "fine"), ctypes.byref(grid), stream_ptr()),
          "lsf_prolong_repeat")
    return out


def downsample2x_linear(fine, channels):
    """3-D, fine: z,y,x (channels == 1) or z,y,x,4"""
    spatial = tuple(fine.shape) if channels == 1 else tuple(fine.shape[:-1])
    if len(spatial) != 3:
        raise NotImplementedError("Cases other than 3D not yet implemented")
    if any(s % 2 for s in spatial):
        raise ValueError("Each field dimension must be evenly divisible by 2.")
    grid = make_grid(spatial)
    out = torch.empty(tuple(s // 2 for s in spatial) + (() if channels == 1 else (channels,)), dtype=torch.float32,
                      device=fine.device)
    check(lib.lsf_downsample2x_linear(_ptr(fine, n_voxels(grid) * channels, "fine"), _ptr(out, out.numel(), "coarse"),
                                      ctypes.byref(grid), channels, stream_ptr()), "lsf_downsample2x_linear")
    return out


def upsample2x_linear(coarse, channels=1):
    spatial = tuple(coarse.shape) if channels == 1 else tuple(coarse.shape[:-1])
    if len(spatial) != 3:
        raise NotImplementedError("Cases other than 3D not yet implemented")
    fine_spatial = tuple(2 * s for s in spatial)
    grid = make_grid(fine_spatial)
    out = torch.empty(fine_spatial + (() if channels == 1 else (channels,)), dtype=torch.float32, device=coarse.device)
    check(lib.lsf_upsample2x_linear(_ptr(coarse, coarse.numel(), "coarse"), _ptr(out, out.numel(), "fine"),
                                    ctypes.byref(grid), channels, stream_ptr()), "lsf_upsample2x_linear")
    return out


# ---------------------------------------------------------------------------------------------- a9/a10
LISTED_TAP_COUNTS = (3, 5, 7, 9)  # lsf_convolve_axis_listed


def convolve_axis(src, dst, zero_mask_source, grid, axis, taps, gate=None, band=None):
    """one pass of the separable filter; band: an LSF_BAND_ALL list -- the zero-preserving pass at its voxels only"""
    taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float64))
    if taps.ndim != 1 or not (1 <= taps.size <= _lib.MAX_KERNEL_TAPS):
        raise ValueError("kernel must be 1-D with 1..%d taps" % _lib.MAX_KERNEL_TAPS)
    length = (grid.nx, grid.ny, grid.nz)[axis]
    if length < taps.size:
        # the reference cannot do this either: np.convolve(..., 'same') returns max(M, N) samples
        raise ValueError("cannot convolve a field of extent %d with a %d-tap kernel" % (length, taps.size))
    planes = src.shape[0]
    n = n_voxels(grid) * planes
    if band is not None:
        check(lib.lsf_convolve_axis_listed(_ptr(src, n, "conv src"), _ptr(dst, n, "conv dst"),
                                           _ptr(zero_mask_source, n, "zero mask"), ctypes.byref(grid), planes, axis,
                                           taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size,
                                           _gate_ref(gate), band.pointer, band.count, stream_ptr()),
              "lsf_convolve_axis_listed")
        return
    check(lib.lsf_convolve_axis(_ptr(src, n, "conv src"), _ptr(dst, n, "conv dst"),
                                _ptr(zero_mask_source, n, "zero mask", allow_none=True), ctypes.byref(grid),
                                planes, axis, taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size,
                                _gate_ref(gate), stream_ptr()), "lsf_convolve_axis")


XYZ_TAP_COUNTS = (3, 5, 7, 9)


def convolve_xy_ok(grid, taps):
    """x and y pass of a 3-D filter in one launch (lsf_convolve_xy)?"""
    return (grid.dims == 3 and len(taps) in XYZ_TAP_COUNTS and grid.nx % 4 == 0 and min(grid.nx, grid.ny) >= len(taps)
            and ((grid.ny + 15) // 16) * (grid.z_end - grid.z_begin) <= 65535)


def convolve_xy(src, dst, grid, taps, gate=None):
    """dst = the y pass of the x pass of src (3-D, no zero mask), on the grid's z-range"""
    taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float64))
    planes = src.shape[0]
    n = n_voxels(grid) * planes
    check(lib.lsf_convolve_xy(_ptr(src, n, "conv src"), _ptr(dst, n, "conv dst"), ctypes.byref(grid), planes,
                              taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size, _gate_ref(gate),
                              stream_ptr()), "lsf_convolve_xy")


def convolve_axis_update_ok(grid, taps):
    """can the filter's last pass also move the warp (lsf_convolve_axis_update: the register-window pass along z, whose
    launch grid must fit)?  3-D only: the reference's 2-D filter ends with its x pass (math_utils/convolution.py:77-83)"""
    if len(taps) not in XYZ_TAP_COUNTS or grid.dims != 3:
        return False
    slices = grid.z_end - grid.z_begin
    return ((slices + 31) // 32) * ((grid.ny + 3) // 4) <= 65535


def convolve_axis_update(src, dst, warp, rate, grid, axis, taps, gate=None):
    """the last pass of a hierarchical iteration's filter, which also moves the warp: warp -= rate * dst"""
    taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float64))
    length = (grid.nx, grid.ny, grid.nz)[axis]
    if length < taps.size:  # (as convolve_axis)
        raise ValueError("cannot convolve a field of extent %d with a %d-tap kernel" % (length, taps.size))
    planes = src.shape[0]
    n = n_voxels(grid) * planes
    check(lib.lsf_convolve_axis_update(_ptr(src, n, "conv src"), _ptr(dst, n, "conv dst"), _ptr(warp, n, "warp"),
                                       float(rate), ctypes.byref(grid), planes, axis,
                                       taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size, _gate_ref(gate),
                                       stream_ptr()), "lsf_convolve_axis_update")


def convolve_xyz_ok(grid, taps):
    """can lsf_convolve_xyz run this 3-D filter (nx % 4 == 0, 3 / 5 / 7 / 9 taps that fit every axis)?"""
    n = len(taps)
    return (grid.dims == 3 and n in XYZ_TAP_COUNTS and grid.nx % 4 == 0 and min(grid.nx, grid.ny, grid.nz) >= n
            and ((grid.ny + 15) // 16) * ((grid.z_end - grid.z_begin + 31) // 32) <= 65535)


def convolve_xyz(src, dst, grid, taps, gate=None, warp=None, rate=0.0):
    """the x, y and z passes of convolve_axis (no zero mask) in one launch (lsf_convolve_xyz): same result, one read
    and one write of the field instead of three; warp: also warp -= rate * dst (the hierarchical update)"""
    taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float64))
    if not convolve_xyz_ok(grid, taps):
        raise ValueError("lsf_convolve_xyz cannot run this grid / kernel; use three convolve_axis passes")
    planes = src.shape[0]
    n = n_voxels(grid) * planes
    check(lib.lsf_convolve_xyz(_ptr(src, n, "conv src"), _ptr(dst, n, "conv dst"),
                               _ptr(warp, n, "warp", allow_none=True), float(rate), ctypes.byref(grid), planes,
                               taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size, _gate_ref(gate),
                               stream_ptr()), "lsf_convolve_xyz")
