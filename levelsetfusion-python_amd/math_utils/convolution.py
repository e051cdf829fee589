"""Separable convolution of 2-D (H,W,2) and 3-D (D,H,W,3) vector fields on the GPU
(reference: math_utils/convolution.py:70-132).  In place, like the reference; pass order 2-D: y then x,
3-D: x, y, z; true convolution, zero padded, float64 accumulation rounded to float32 after every pass."""
import numpy as np

from .. import device as dev
from ..engine_common import _conv_axis_order, as_device_field

# the (size 7, strength 0.1) filter as float64 -- numerical data of math_utils/convolution.py:20-26
sobolev_kernel_1d = np.array([2.995900285895913839e-04, 4.410949535667896271e-03, 6.571318954229354858e-02,
                              9.956527948379516602e-01, 6.571318954229354858e-02, 4.410949535667896271e-03,
                              2.995900285895913839e-04])


def _run_passes(vector_field, kernel, axes, preserve_zeros):
    v = as_device_field(vector_field)
    dims = v.dim() - 1
    if v.shape[-1] != dims or dims not in (2, 3):
        raise ValueError("Can only process tensors with 3 dimensions (where last dimension is 2) or "
                         "tensors with 4 dimensions (where last dimension is 3), i.e. 2D & 3D vector fields")
    grid = dev.make_grid(v.shape[:-1])
    src = dev.deinterleave(v, dims)
    mask = src if preserve_zeros else None
    bufs = [src.clone() if preserve_zeros else src, None]
    cur = bufs[0]
    for axis in axes:
        out = cur.new_empty(cur.shape)
        dev.convolve_axis(cur, out, mask, grid, axis, kernel)
        cur = out
    np.copyto(vector_field, dev.interleave(cur).cpu().numpy())
    return vector_field


def convolve_with_kernel(vector_field, kernel=sobolev_kernel_1d, print_focus_coord_info=False):
    return _run_passes(vector_field, kernel, _conv_axis_order(vector_field.ndim - 1), False)


def convolve_with_kernel_preserve_zeros(vector_field, kernel=sobolev_kernel_1d, print_focus_coord_info=False):
    """components with |v| < 1e-6 in the INPUT are forced back to zero after every pass
    (math_utils/convolution.py:114-132)"""
    return _run_passes(vector_field, kernel, _conv_axis_order(vector_field.ndim - 1), True)


def convolve_with_kernel_x(vector_field, kernel):
    return _run_passes(vector_field, kernel, [0], False)


def convolve_with_kernel_y(vector_field, kernel):
    return _run_passes(vector_field, kernel, [1], False)
