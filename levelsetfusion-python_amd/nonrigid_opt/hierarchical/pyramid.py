"""Scalar field pyramids built on the GPU (reference: nonrigid_opt/hierarchical/pyramid.py:28-56):
repeated 2x2 (2x2x2) block means, levels stored coarsest first, same ValueErrors."""
from ... import device as dev
from ...engine import as_device_field, pyramid_level_count


class _ScalarFieldPyramid:
    DIMS = 2

    def __init__(self, field, maximum_chunk_size=8):
        if len(field.shape) != self.DIMS:
            raise ValueError("expected a %d-D field" % self.DIMS)
        n_levels = pyramid_level_count(field.shape, maximum_chunk_size)
        device_levels = [as_device_field(field).clone()]
        for _ in range(1, n_levels):
            device_levels.append(dev.restrict_mean(device_levels[-1], 1))
        device_levels.reverse()
        self.device_levels = device_levels
        self.levels = [lvl.cpu().numpy() for lvl in device_levels]


class ScalarFieldPyramid2d(_ScalarFieldPyramid):
    DIMS = 2


class ScalarFieldPyramid3d(_ScalarFieldPyramid):
    DIMS = 3
