"""3-D linear 2x resampling on the GPU (reference prototype: math_utils/resampling.py:29-126, the Python twin of the C++
optimizer's ResamplingStrategy.LINEAR).  numpy in, numpy out.  Computed in float32 (prolongation, as the reference does
for float32 input) / accumulated in float64 and stored as float32 (restriction); the reference returns float64 arrays."""
import numpy as np

from .. import device as dev
from ..engine import as_device_field


def upsample2x_linear(field):
    """0.75 / 0.25 trilinear prolongation with edge padding (resampling.py:29-80)"""
    if len(field.shape) != 3:
        raise NotImplementedError("Cases other than 3D not yet implemented")
    return dev.upsample2x_linear(as_device_field(field)).cpu().numpy()


def downsample2x_linear(field):
    """4x4x4-window restriction with the reference's literal weights, edge padding (resampling.py:83-126)"""
    if len(field.shape) != 3:
        raise NotImplementedError("Cases other than 3D not yet implemented")
    if field.shape[0] % 2 != 0 or field.shape[1] % 2 != 0 or field.shape[2] % 2 != 0:
        raise ValueError("Each field dimension must be evenly divisible by 2.")
    return dev.downsample2x_linear(as_device_field(field), 1).cpu().numpy()
