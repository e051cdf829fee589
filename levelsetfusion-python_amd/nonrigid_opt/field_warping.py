"""Resampling of scalar fields under a warp, on the GPU (reference: nonrigid_opt/field_warping.py:67-151).
numpy in, numpy out, same argument order and the same in-place mutation of warp_field / gradient_field in
warp_field_advanced.  2-D and 3-D."""
import numpy as np

from .. import device as dev
from ..engine import as_device_field


def _planar_warp(vector_field, dims):
    v = as_device_field(vector_field)
    if v.dim() != dims + 1 or v.shape[-1] != dims:
        raise ValueError("expected a vector field of shape %s + (%d,)" % ("field.shape", dims))
    return dev.deinterleave(v, dims)


def warp_field(field, vector_field):
    """D-linear lookup of `field` at p + vector_field[p]; out-of-bounds taps read 1 (field_warping.py:67-85)"""
    f = as_device_field(field)
    return dev.warp_field(f, _planar_warp(vector_field, f.dim()), 1.0).cpu().numpy()


def warp_field_replacement(field, warps, replacement):
    """as warp_field, out-of-bounds taps read `replacement` (field_warping.py:88-109)"""
    f = as_device_field(field)
    return dev.warp_field(f, _planar_warp(warps, f.dim()), float(replacement)).cpu().numpy()


def warp_field_advanced(canonical_field, warped_live_field, warp_field, gradient_field, band_union_only=False,
                        known_values_only=False, substitute_original=False):
    """truncation-aware re-warp (field_warping.py:112-151): returns the new live field; where the resampled value
    snaps to +-1, warp_field[p] and gradient_field[p] are zeroed IN PLACE (numpy arrays)."""
    live = as_device_field(warped_live_field)
    dims = live.dim()
    warp_p = _planar_warp(warp_field, dims)
    grad_p = _planar_warp(gradient_field, dims) if gradient_field is not None else None
    flags = (1 if band_union_only else 0) | (2 if known_values_only else 0) | (4 if substitute_original else 0)
    new_live = dev.warp_field_advanced(as_device_field(canonical_field), live, warp_p, grad_p, flags)
    np.copyto(warp_field, dev.interleave(warp_p).cpu().numpy())
    if gradient_field is not None:
        np.copyto(gradient_field, dev.interleave(grad_p).cpu().numpy())
    return new_live.cpu().numpy()
