"""Hard-coded piecewise-linear ("orthographic") 2-D TSDF pairs -- the input of BASELINE config 1 and of the reference's
single-frame experiment and `run_slavcheva_optimizer2d.py` (reference: tsdf/generation.py:238-353, utils/point2d.py).

Host numpy by design: a one-off fill of a 128 x 128 field, column by column along a polyline (the optimizers upload the
result once).  One polyline segment is filled with array operations; the arithmetic keeps the reference's types: the
surface points are float32 (they come out of a float32 array, `+= offset` and `+ 5.0` stay float32 under numpy 2's
promotion rules), voxel coordinates are Python ints, so every intermediate is a float32 operation.
"""
import numpy as np


class Point2d:
    """what the generators read from utils.point2d.Point2d: .x, .y"""

    def __init__(self, x=0.0, y=0.0, coordinates=None):
        if coordinates is not None:
            x, y = coordinates[0], coordinates[1]
        self.x = x
        self.y = y

    def __repr__(self):
        return "[{:>03.2f},{:>03.2f}]".format(float(self.x), float(self.y))


def _xy(point):
    if hasattr(point, "x"):
        return np.float32(point.x), np.float32(point.y)
    return np.float32(point[0]), np.float32(point[1])


def add_surface_to_2d_tsdf_field_sample(field, consecutive_surface_points, narrow_band_width_voxels=20,
                                        back_cutoff_voxels=np.inf):
    """Writes, into every column x between consecutive surface points, 1 above the narrow band, the truncated signed
    distance `(surface_y - y) / half_width` inside it and -1 behind it (unless `back_cutoff_voxels` ends the band
    early: the rows behind the cut-off keep their value).  In place; returns `field` (tsdf/generation.py:238-265).
    Raises ValueError when the surface comes closer than the band width to row 0 and IndexError when a column lies
    outside the field, as the reference's loops do (columns before the offending one are already written)."""
    half_width = narrow_band_width_voxels // 2
    back = min(half_width, back_cutoff_voxels)
    rows = np.arange(field.shape[0])[:, None]
    for point_a, point_b in zip(consecutive_surface_points[:-1], consecutive_surface_points[1:]):
        ax, ay = _xy(point_a)
        bx, by = _xy(point_b)
        x = np.arange(int(ax), int(bx))
        if x.size == 0:
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = (x.astype(np.float32) - ax) / (bx - ax)
        surface_y = ay * (np.float32(1.0) - ratio) + by * ratio  # float32 throughout
        too_close = surface_y - np.float32(narrow_band_width_voxels) < 0
        outside = (x >= field.shape[1]) | (x < -field.shape[1])
        # a band that runs past the last row: the reference writes the rows that exist, then raises
        past_last_row = (surface_y + np.float32(back) + np.float32(1)).astype(np.int64) > field.shape[0]
        stop = np.flatnonzero(too_close | outside | past_last_row)
        n_ok = x.size
        if stop.size:
            n_ok = int(stop[0]) + (0 if (too_close | outside)[stop[0]] else 1)
        if n_ok:
            xs, sy = x[:n_ok], surface_y[:n_ok]
            start = (sy - np.float32(half_width)).astype(np.int64)  # int() truncates; the operands are positive
            end = (sy + np.float32(back) + np.float32(1)).astype(np.int64)
            distance = np.clip((sy[None, :] - rows.astype(np.float32)) / np.float32(half_width), -1.0, 1.0)
            column = field[:, xs]
            column = np.where(rows < start[None, :], np.float32(1.0), column)
            column = np.where((rows >= start[None, :]) & (rows < end[None, :]), distance.astype(np.float32), column)
            fill_back = (end < field.shape[0]) & (end < back_cutoff_voxels)
            column = np.where((rows >= end[None, :]) & fill_back[None, :], np.float32(-1.0), column)
            field[:, xs] = column
        if stop.size:
            if too_close[stop[0]]:
                raise ValueError("Surface is too close to 0 in the y dimension for a full narrow band representation")
            if outside[stop[0]]:
                raise IndexError("index %d is out of bounds for axis 1 with size %d" % (x[stop[0]], field.shape[1]))
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (field.shape[0], field.shape[0]))
    return field


def generate_sample_orthographic_2d_tsdf_field(consecutive_surface_points, size, narrow_band_width_voxels=20,
                                               back_cutoff_voxels=np.inf, default_value=1):
    """a (size, size) float32 field of `default_value` with one polyline surface (tsdf/generation.py:269-279);
    back_cutoff_voxels=3 mimics SobolevFusion's eta parameter"""
    field = np.full((size, size), default_value, dtype=np.float32)
    return add_surface_to_2d_tsdf_field_sample(field, consecutive_surface_points, narrow_band_width_voxels,
                                               back_cutoff_voxels)


# the reference's polyline (tsdf/generation.py:289-305) -- data, (x, y) voxel coordinates
_SURFACE_POINTS = np.array([[9, 56], [14, 66], [23, 72], [35, 72], [44, 65], [54, 60], [63, 60], [69, 64], [76, 71],
                            [84, 73], [91, 72], [106, 63], [109, 57]], dtype=np.float32)
_SURFACE_POINTS_EXTRA = np.array([[32, 65], [36, 65], [41, 61]], dtype=np.float32)
_SURFACE_OFFSET = -0.23          # "unrealistic to expect even values", tsdf/generation.py:288
_CANONICAL_SHIFT = 5.0           # canonical = live surface shifted by +5 rows, tsdf/generation.py:328


def generate_initial_orthographic_2d_tsdf_fields(field_size=128, narrow_band_width_voxels=20, mimic_eta=False,
                                                 live_smoothing_kernel_size=0, canonical_smoothing_kernel_size=0,
                                                 default_value=1):
    """(live_field, canonical_field): the reference's hand-made 2-D pair -- a polyline surface plus a three-point
    detail over columns 32..40, canonical = the same surface 5 rows further down, optionally cut off 3 voxels behind
    the surface (`mimic_eta`).  The field must hold column 108, i.e. field_size >= 109 (smaller sizes raise IndexError
    as the reference does).  The reference's two Gaussian-smoothing arguments refer to names its module never defines
    (tsdf/generation.py:345-351: NameError for any size > 0); the same error is raised here."""
    offset = np.float32(_SURFACE_OFFSET)
    live_points = [Point2d(x, y + offset) for x, y in _SURFACE_POINTS]
    live_extra = [Point2d(x, y + offset) for x, y in _SURFACE_POINTS_EXTRA]
    live_field = generate_sample_orthographic_2d_tsdf_field(live_points, field_size, narrow_band_width_voxels,
                                                            default_value=default_value)
    live_field = add_surface_to_2d_tsdf_field_sample(live_field, live_extra, narrow_band_width_voxels)
    shift = np.float32(_CANONICAL_SHIFT)
    canonical_points = [Point2d(p.x, p.y + shift) for p in live_points]
    canonical_extra = [Point2d(p.x, p.y + shift) for p in live_extra]
    back_cutoff_voxels = 3 if mimic_eta else np.inf
    canonical_field = generate_sample_orthographic_2d_tsdf_field(canonical_points, field_size,
                                                                 narrow_band_width_voxels, back_cutoff_voxels,
                                                                 default_value)
    canonical_field = add_surface_to_2d_tsdf_field_sample(canonical_field, canonical_extra, narrow_band_width_voxels,
                                                          back_cutoff_voxels)
    if live_smoothing_kernel_size > 0 or canonical_smoothing_kernel_size > 0:
        raise NameError("name 'IGNORE_OPENCV' is not defined (the reference's smoothing branch, "
                        "tsdf/generation.py:345-351, cannot run either)")
    return live_field, canonical_field
