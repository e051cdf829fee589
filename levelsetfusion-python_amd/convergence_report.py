"""Convergence report structures.  The reference builds these in its C++ extension
(cpp.build_warp_delta_statistics_2d / build_tsdf_difference_statistics_2d / ConvergenceReport2d,
nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:393-404); field names follow
run_hierarchical_optimizer3d_multipair.py:112-130.  The reductions run on the GPU
(lsf_warp_statistics / lsf_tsdf_difference_statistics)."""
import math

import numpy as np

from . import device as dev


def _close(a, b, tol=1e-6):
    return abs(a - b) <= tol


class Location(tuple):
    """voxel location (x, y[, z]); compares like a tuple and has the .x/.y/.z attributes of the reference's
    Vector2i / Vector3i"""

    def __new__(cls, *coords):
        if len(coords) == 1 and not isinstance(coords[0], (int, np.integer)):
            coords = tuple(coords[0])
        return super().__new__(cls, tuple(int(c) for c in coords))

    x = property(lambda self: self[0])
    y = property(lambda self: self[1])
    z = property(lambda self: self[2] if len(self) > 2 else 0)


class WarpDeltaStatistics:
    def __init__(self, ratio_above_min_threshold=0.0, length_min=0.0, length_max=0.0, length_mean=0.0,
                 length_standard_deviation=0.0, longest_warp_location=(0, 0), is_largest_below_min_threshold=False,
                 is_largest_above_max_threshold=False):
        self.ratio_above_min_threshold = ratio_above_min_threshold
        self.length_min = length_min
        self.length_max = length_max
        self.length_mean = length_mean
        self.length_standard_deviation = length_standard_deviation
        self.longest_warp_location = Location(longest_warp_location)
        self.is_largest_below_min_threshold = is_largest_below_min_threshold
        self.is_largest_above_max_threshold = is_largest_above_max_threshold

    def __eq__(self, other):
        return (_close(self.ratio_above_min_threshold, other.ratio_above_min_threshold)
                and _close(self.length_min, other.length_min) and _close(self.length_max, other.length_max)
                and _close(self.length_mean, other.length_mean)
                and _close(self.length_standard_deviation, other.length_standard_deviation)
                and self.longest_warp_location == other.longest_warp_location
                and self.is_largest_below_min_threshold == other.is_largest_below_min_threshold
                and self.is_largest_above_max_threshold == other.is_largest_above_max_threshold)

    def __repr__(self):
        return "WarpDeltaStatistics(%r)" % (self.__dict__,)


class TsdfDifferenceStatistics:
    def __init__(self, difference_min=0.0, difference_max=0.0, difference_mean=0.0,
                 difference_standard_deviation=0.0, biggest_difference_location=(0, 0)):
        self.difference_min = difference_min
        self.difference_max = difference_max
        self.difference_mean = difference_mean
        self.difference_standard_deviation = difference_standard_deviation
        self.biggest_difference_location = Location(biggest_difference_location)

    def __eq__(self, other):
        return (_close(self.difference_min, other.difference_min) and _close(self.difference_max, other.difference_max)
                and _close(self.difference_mean, other.difference_mean)
                and _close(self.difference_standard_deviation, other.difference_standard_deviation)
                and self.biggest_difference_location == other.biggest_difference_location)

    def __repr__(self):
        return "TsdfDifferenceStatistics(%r)" % (self.__dict__,)


class ConvergenceReport:
    def __init__(self, iteration_count=0, iteration_limit_reached=False, warp_delta_statistics=None,
                 tsdf_difference_statistics=None):
        self.iteration_count = iteration_count
        self.iteration_limit_reached = iteration_limit_reached
        self.warp_delta_statistics = warp_delta_statistics or WarpDeltaStatistics()
        self.tsdf_difference_statistics = tsdf_difference_statistics or TsdfDifferenceStatistics()

    def __eq__(self, other):
        return (self.iteration_count == other.iteration_count
                and self.iteration_limit_reached == other.iteration_limit_reached
                and self.warp_delta_statistics == other.warp_delta_statistics
                and self.tsdf_difference_statistics == other.tsdf_difference_statistics)

    def __repr__(self):
        return "ConvergenceReport(%d, %r, %r, %r)" % (self.iteration_count, self.iteration_limit_reached,
                                                      self.warp_delta_statistics, self.tsdf_difference_statistics)


ConvergenceReport2d = ConvergenceReport
ConvergenceReport3d = ConvergenceReport


def _location(linear_index, shape):
    """linear voxel index -> (x, y[, z]) as the reference's Vector2i/3i order them"""
    if linear_index < 0:
        return tuple(0 for _ in shape)
    idx = np.unravel_index(int(linear_index), shape)
    return tuple(int(i) for i in idx[::-1])


def build_warp_delta_statistics(warp_planar, canonical, live, lower_threshold, upper_threshold):
    """statistics of |warp| over the narrow-band union (semantics recovered from the known answer at
    tests/test_slavcheva_optimizer.py:141-145: ratio / max / mean / population std / arg-max over band-union
    voxels, length_min reported as 0)."""
    raw = dev.warp_statistics(warp_planar, canonical, live, lower_threshold).cpu().numpy()
    return warp_delta_statistics_from_raw(raw, tuple(live.shape), lower_threshold, upper_threshold)


def warp_delta_statistics_from_raw(raw, shape, lower_threshold, upper_threshold):
    """raw: the 8 doubles of lsf_warp_statistics (or lsf_state_finalize's statistics16[0:8])"""
    count, above, mx, s1, s2, arg = raw[0], raw[1], raw[2], raw[3], raw[4], raw[5]
    if count == 0:
        return WarpDeltaStatistics()
    mean = s1 / count
    var = max(s2 / count - mean * mean, 0.0)
    return WarpDeltaStatistics(above / count, 0.0, float(mx), float(mean), math.sqrt(var),
                               _location(arg, shape), bool(mx < lower_threshold), bool(mx > upper_threshold))


def build_tsdf_difference_statistics(canonical, live):
    """statistics of |canonical - live| over ALL voxels (same known answer)."""
    raw = dev.tsdf_difference_statistics(canonical, live).cpu().numpy()
    return tsdf_difference_statistics_from_raw(raw, tuple(live.shape))


def tsdf_difference_statistics_from_raw(raw, shape):
    """raw: the 8 doubles of lsf_tsdf_difference_statistics (or lsf_state_finalize's statistics16[8:16])"""
    count, mn, mx, s1, s2, arg = raw[0], raw[1], raw[2], raw[3], raw[4], raw[5]
    mean = s1 / count
    var = max(s2 / count - mean * mean, 0.0)
    return TsdfDifferenceStatistics(float(mn), float(mx), float(mean), math.sqrt(var), _location(arg, shape))
