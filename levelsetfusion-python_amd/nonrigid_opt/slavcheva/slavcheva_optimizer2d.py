"""SlavchevaOptimizer2d -- drop-in for nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:74-411 of the reference
(KillingFusion / SobolevFusion style optimizer): same constructor keywords and enums, same
optimize(live_field, canonical_field) convention (argument order REVERSED with respect to the hierarchical
optimizer; live_field is warped IN PLACE and returned), same .log / .get_convergence_report() / .gradient_field
surface.  Differences, all deliberate: no video/plot writers are opened and nothing is printed unless
verbose=True (the reference prints every iteration); both compute methods run the same HIP kernels and differ
only where the reference's two code paths differ arithmetically."""
import os
from enum import Enum

import numpy as np
import torch

from ... import _lib, device as dev
from ...convergence_report import (ConvergenceReport, tsdf_difference_statistics_from_raw,
                                   warp_delta_statistics_from_raw)
from ...engine import SlavchevaEngine, as_device_field
from .data_term import DataTermMethod
from .smoothing_term import SmoothingTermMethod


class AdaptiveLearningRateMethod(Enum):
    NONE = 0
    RMS_PROP = 1   # the reference only allocates an unused buffer for it (slavcheva_optimizer2d.py:348-350)


class ComputeMethod(Enum):
    DIRECT = 0
    VECTORIZED = 1


class OptimizationLog:
    """slavcheva_optimizer2d.py's log of a call: four lists with one entry per iteration, the locations of the longest
    updates and the convergence report.  With a `source` (the engine's log of the call) the lists are taken from it when
    they are first read -- what the call leaves on the host is looked at by whoever wants it, not converted behind the
    call's last synchronisation"""
    _LISTS = {"data_energies": "data_energies", "smoothing_energies": "smoothing_energies",
              "level_set_energies": "level_set_energies", "max_warps": "max_warps"}

    def __init__(self, source=None):
        self._source = source
        self._convergence_report = ConvergenceReport()
        self._max_warp_locations = []
        if source is None:
            self.data_energies = []
            self.smoothing_energies = []
            self.level_set_energies = []
            self.max_warps = []

    def __getattr__(self, name):  # only reached for attributes that have not been set
        key = OptimizationLog._LISTS.get(name)
        source = self.__dict__.get("_source")
        if key is None or source is None:
            raise AttributeError(name)
        value = source[key]
        setattr(self, name, value)
        return value

    @property
    def convergence_report(self):
        """slavcheva_optimizer2d.py:393-404; built from the finalize pass's sixteen sums when somebody looks"""
        if callable(self._convergence_report):
            self._convergence_report = self._convergence_report()
        return self._convergence_report

    @convergence_report.setter
    def convergence_report(self, value):
        self._convergence_report = value

    @property
    def max_warp_locations(self):
        """(x, y[, z]) of every iteration's longest update; given as flat voxel indices + the field's shape, they are
        unravelled when somebody looks (a 50-iteration call's list costs 10 us of host time behind the call's last
        synchronisation, where the card waits for the next call)"""
        pending = self._max_warp_locations
        if isinstance(pending, tuple):
            indices, shape = pending
            coords = np.unravel_index(np.asarray(indices, dtype=np.int64), shape)
            self._max_warp_locations = pending = list(zip(*(c.tolist() for c in coords[::-1])))
        return pending

    @max_warp_locations.setter
    def max_warp_locations(self, value):
        self._max_warp_locations = value


class VoxelLog:
    """slavcheva_optimizer2d.py:48-55: what one voxel of the focus neighbourhood went through, iteration by iteration"""

    def __init__(self):
        self.warp_magnitudes = []
        self.sdf_values = []
        self.canonical_sdf = 0.0

    def __repr__(self):
        return str(self.warp_magnitudes) + "; " + str(self.sdf_values)


class _SlavchevaOptimizerBase:
    DIMS = 2

    def __init__(self, out_path="out2D", field_size=128, default_value=1.0, compute_method=ComputeMethod.DIRECT,
                 level_set_term_enabled=False, sobolev_smoothing_enabled=False,
                 data_term_method=DataTermMethod.BASIC, smoothing_term_method=SmoothingTermMethod.TIKHONOV,
                 adaptive_learning_rate_method=AdaptiveLearningRateMethod.NONE, gradient_descent_rate=0.1,
                 data_term_weight=1.0, smoothing_term_weight=0.2, isomorphic_enforcement_factor=0.1,
                 level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.1,
                 maximum_warp_length_upper_threshold=10000, max_iterations=100, min_iterations=1,
                 sobolev_kernel=None, visualization_settings=None, enable_convergence_status_logging=True,
                 verbose=False, check_interval=32, comm=None, engine_options=None, focus_coordinates=None):
        self.visualization_settings = visualization_settings  # accepted, unused
        self.field_size = field_size
        self.out_path = out_path
        if out_path and not os.path.exists(out_path):  # slavcheva_optimizer2d.py:112-113
            os.makedirs(out_path, exist_ok=True)
        self.total_data_energy = 0.
        self.total_smoothing_energy = 0.
        self.total_level_set_energy = 0.
        self.compute_method = compute_method
        self.level_set_term_enabled = level_set_term_enabled
        self.sobolev_smoothing_enabled = sobolev_smoothing_enabled
        self.gradient_descent_rate = gradient_descent_rate
        self.data_term_weight = data_term_weight
        self.smoothing_term_weight = smoothing_term_weight
        self.isomorphic_enforcement_factor = isomorphic_enforcement_factor
        self.level_set_term_weight = level_set_term_weight
        self.maximum_warp_length_lower_threshold = maximum_warp_length_lower_threshold
        self.maximum_warp_length_upper_threshold = maximum_warp_length_upper_threshold
        self.max_iterations = max_iterations
        self.min_iterations = min_iterations
        self.sobolev_kernel = sobolev_kernel
        self.data_term_method = data_term_method
        self.smoothing_term_method = smoothing_term_method
        self.adaptive_learning_rate_method = adaptive_learning_rate_method
        self.default_value = default_value
        self.enable_convergence_status_logging = enable_convergence_status_logging
        self.verbose = verbose
        self.log = None
        self._warp_field = None
        self.focus_coordinates = focus_coordinates
        self.focus_neighborhood_log = None
        self._engine = SlavchevaEngine(
            direct=compute_method == ComputeMethod.DIRECT, level_set_term_enabled=level_set_term_enabled,
            sobolev_smoothing_enabled=sobolev_smoothing_enabled,
            data_term_method=_lib.DATA_THRESHOLDED_FDM if data_term_method == DataTermMethod.THRESHOLDED_FDM
            else _lib.DATA_BASIC,
            smoothing_term_method=_lib.SMOOTHING_KILLING if smoothing_term_method == SmoothingTermMethod.KILLING
            else _lib.SMOOTHING_TIKHONOV,
            gradient_descent_rate=gradient_descent_rate, data_term_weight=data_term_weight,
            smoothing_term_weight=smoothing_term_weight, isomorphic_enforcement_factor=isomorphic_enforcement_factor,
            level_set_term_weight=level_set_term_weight, lower_threshold=maximum_warp_length_lower_threshold,
            upper_threshold=maximum_warp_length_upper_threshold, max_iterations=max_iterations,
            min_iterations=min_iterations,
            sobolev_kernel=None if sobolev_kernel is None else np.asarray(sobolev_kernel, dtype=np.float64),
            check_interval=check_interval, comm=comm, options=engine_options)

    def _run_checks(self, live_field, canonical_field):
        # slavcheva_optimizer2d.py:157-161,339: equal shapes, square/cubic, side == field_size
        shape = tuple(live_field.shape)
        if shape != tuple(canonical_field.shape) or len(shape) != self.DIMS \
                or any(s != self.field_size for s in shape):
            raise ValueError("warp field, warped live field, and canonical field all need to be arrays of the same "
                             "size (field_size = %d on every side)." % self.field_size)

    @property
    def engine(self):
        """the SlavchevaEngine behind this optimizer: its knobs (engine_options.SLAVCHEVA_DEFAULTS, also settable through
        the constructor's `engine_options=dict(...)`) and `last_call`, the report of what the last call took"""
        return self._engine

    def _focus_neighbourhood(self, shape):
        """slavcheva_optimizer2d.py:422-430: the voxels within one step of the focus coordinate (x, y[, z]) that lie
        inside the field, as (x, y[, z]) keys in the reference's order (x fastest)"""
        focus = tuple(int(c) for c in self.focus_coordinates)
        if len(focus) != self.DIMS:
            raise ValueError("focus_coordinates must be (x, y%s)" % (", z" if self.DIMS == 3 else ""))
        keys = [()]
        for axis in range(self.DIMS):  # x first: every later axis becomes the slower one
            extent = shape[self.DIMS - 1 - axis]
            keys = [k + (c,) for c in range(focus[axis] - 1, focus[axis] + 2) if 0 <= c < extent for k in keys]
        return keys

    @property
    def iteration_hook(self):
        """opt-in call-back f(level = 0, iteration, warp_field, gradient_field, max_warp) after every iteration, where the
        reference writes its per-iteration visualisations (slavcheva_optimizer2d.py:387-388); device tensors [..., D].
        None (default) costs nothing; with a hook every iteration is synchronised (and the fused kernel's gradient, which
        it does not store, is recomputed by the unfused kernels)."""
        return self._engine.iteration_hook

    @iteration_hook.setter
    def iteration_hook(self, hook):
        self._engine.iteration_hook = hook

    @property
    def warp_field(self):
        """the last call's final warp field [..., D] (numpy for numpy inputs, a device tensor for device inputs).  The
        reference's optimize() keeps its warp field to itself (slavcheva_optimizer2d.py:332-408: a local); here it is an
        attribute that is BUILT WHEN READ -- the iterations keep the warp packed with the live field, and a dense
        [..., D] copy costs a pass over the volume nobody may ever look at"""
        w = self._warp_field
        if callable(w):
            w = self._warp_field = w()
        return w

    @warp_field.setter
    def warp_field(self, value):
        self._warp_field = value

    @property
    def gradient_field(self):
        """gradient of the last iteration, interleaved numpy array (reference attribute of the same name)"""
        g = self._engine.gradient_field()
        return None if g is None else dev.interleave(g).cpu().numpy()

    def optimize(self, live_field, canonical_field):
        on_device = isinstance(live_field, torch.Tensor) and live_field.is_cuda
        if self._engine.comm is None or not self._engine.comm.active:
            self._run_checks(live_field, canonical_field)
        live = as_device_field(live_field)
        canonical = as_device_field(canonical_field)
        # one pass over the final state: the live field back into the caller's array (in place, like np.copyto at
        # :230,:328), the warp in the API layout, and the convergence statistics (:393-404)
        want_report = self.enable_convergence_status_logging and \
            (self._engine.comm is None or not self._engine.comm.active)
        finalize_args = (live_field if on_device else None, self.maximum_warp_length_lower_threshold, want_report)
        # the reference traces the 3 x 3 voxels around its module-global focus coordinate on every call (:336-337,
        # utils/sampling.py:27); here the trace is asked for per optimizer (`focus_coordinates=(x, y[, z])`): a traced call
        # is synchronised every iteration
        keys = None if self.focus_coordinates is None else self._focus_neighbourhood(tuple(live.shape))
        self._engine.focus_voxels = None if keys is None else [k[::-1] for k in keys]
        self.focus_neighborhood_log = None
        outcome = self._engine.optimize(live, canonical, finalize=finalize_args)
        if keys is not None:
            trace = self._engine.focus_trace
            self.focus_neighborhood_log = {}
            for v, key in enumerate(keys):
                entry = self.focus_neighborhood_log[key] = VoxelLog()
                entry.canonical_sdf = trace["canonical"][v]  # :353-354
                entry.warp_magnitudes = [w[v] for w in trace["warp"]]  # :319-322
                entry.sdf_values = [s[v] for s in trace["sdf"]]
        eng_log = self._engine.log
        self.log = OptimizationLog(eng_log)
        if self._engine.comm is None or not self._engine.comm.active:
            self.log.max_warp_locations = (eng_log["max_warp_indices"], tuple(live.shape))  # unravelled when read
        else:
            self.log.max_warp_locations = eng_log["max_warp_indices"]
        n = self._engine.iteration_count
        self._totals_from_log = n > 0  # total_data_energy / _smoothing_ / _level_set_: the last iteration's, when read
        if self.verbose:
            for i in range(n):
                print("[Iteration %d done], data energy: %f; smoothing energy: %f; level set energy: %f; max warp: %f"
                      % (i, self.log.data_energies[i], self.log.smoothing_energies[i],
                         self.log.level_set_energies[i], self.log.max_warps[i]))
        final_live, warp, raw = outcome.finalize(*finalize_args)
        if want_report:
            shape = tuple(live.shape)
            lower, upper, limit = self.maximum_warp_length_lower_threshold, self.maximum_warp_length_upper_threshold, \
                self.max_iterations

            def report():
                ws = warp_delta_statistics_from_raw(raw[:8], shape, lower, upper)
                ds = tsdf_difference_statistics_from_raw(raw[8:], shape)
                return ConvergenceReport(n, n >= limit, ws, ds)
            self.log.convergence_report = report  # evaluated on first access
        if on_device:
            self.warp_field = warp  # a tensor, or a callable that builds it on first access (the warp_field property)
        else:
            np.copyto(live_field, final_live.cpu().numpy())
            self.warp_field = (lambda: warp().cpu().numpy()) if callable(warp) else warp.cpu().numpy()
        return live_field

    def get_convergence_report(self):
        return self.log.convergence_report

    # the energies of the last executed iteration (slavcheva_optimizer2d.py:231-236,:330): read from the call's log
    def _total(self, name, stored):
        if getattr(self, "_totals_from_log", False):
            return getattr(self.log, name)[-1]
        return self.__dict__.get(stored, 0.)

    total_data_energy = property(lambda self: self._total("data_energies", "_total_data"),
                                 lambda self, v: self._set_total("_total_data", v))
    total_smoothing_energy = property(lambda self: self._total("smoothing_energies", "_total_smoothing"),
                                      lambda self, v: self._set_total("_total_smoothing", v))
    total_level_set_energy = property(lambda self: self._total("level_set_energies", "_total_level_set"),
                                      lambda self, v: self._set_total("_total_level_set", v))

    def _set_total(self, stored, value):
        self.__dict__[stored] = value
        self._totals_from_log = False


class SlavchevaOptimizer2d(_SlavchevaOptimizerBase):
    DIMS = 2
