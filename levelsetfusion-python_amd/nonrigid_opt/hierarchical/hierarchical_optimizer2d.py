"""HierarchicalOptimizer2d -- drop-in for nonrigid_opt/hierarchical/hierarchical_optimizer2d.py:62-167 of the
reference: same constructor keywords, same optimize(canonical_field, live_field) -> warp_field (H, W, 2) float32
convention (inputs untouched), same ValueErrors for fields a pyramid cannot be built from.  The work is done by
HIP kernels on the GPU (engine.HierarchicalEngine)."""
import numpy as np
import torch

from ... import device as dev
from ...engine import HierarchicalEngine, as_device_field
from ...convergence_report import (ConvergenceReport, build_tsdf_difference_statistics,
                                   build_warp_delta_statistics)


def _engine_setting(name, convert=None):
    """a constructor setting the reference reads from `self` in every iteration (hierarchical_optimizer2d.py:186-225): kept
    in ONE place, the engine, so that assigning to it after construction takes effect -- and drops the captured HIP
    graphs, which bake rate, threshold, taps and iteration counts in"""
    def get(self):
        return getattr(self._engine, name)

    def put(self, value):
        value = convert(value) if convert is not None and value is not None else value
        # the reference keeps plain attributes and would fail inside its loop on a filter that is switched on without
        # taps (convolve_with_kernel(None)); say so at the assignment instead
        kernel = value if name == "gradient_kernel" else self._engine.gradient_kernel
        enabled = value if name == "gradient_kernel_enabled" else self._engine.gradient_kernel_enabled
        if name in ("gradient_kernel", "gradient_kernel_enabled") and enabled and kernel is None:
            raise ValueError("gradient_kernel_enabled needs a gradient_kernel (assign the kernel first, or disable the "
                             "filter before removing it)")
        setattr(self._engine, name, value)
        self._engine.invalidate_graphs()
    return property(get, put)


class _HierarchicalOptimizerBase:
    DIMS = 2
    maximum_chunk_size = _engine_setting("maximum_chunk_size", int)
    rate = _engine_setting("rate", float)
    data_term_amplifier = _engine_setting("data_term_amplifier", float)
    tikhonov_strength = _engine_setting("tikhonov_strength", float)
    tikhonov_term_enabled = _engine_setting("tikhonov_term_enabled", bool)
    gradient_kernel = _engine_setting("gradient_kernel", lambda k: np.asarray(k, dtype=np.float64))
    gradient_kernel_enabled = _engine_setting("gradient_kernel_enabled", bool)
    maximum_warp_update_threshold = _engine_setting("maximum_warp_update_threshold", float)
    maximum_iteration_count = _engine_setting("maximum_iteration_count", int)

    class VerbosityParameters:
        """stdout verbosity switches (hierarchical_optimizer2d.py:41-60; the 3-D/C++ variant adds the
        mean/std TSDF-difference flags, build_helper.py:127-154)"""

        def __init__(self, print_max_warp_update=False, print_iteration_mean_tsdf_difference=False,
                     print_iteration_std_tsdf_difference=False, print_iteration_data_energy=False,
                     print_iteration_tikhonov_energy=False):
            self.print_max_warp_update = print_max_warp_update
            self.print_iteration_mean_tsdf_difference = print_iteration_mean_tsdf_difference
            self.print_iteration_std_tsdf_difference = print_iteration_std_tsdf_difference
            self.print_iteration_data_energy = print_iteration_data_energy
            self.print_iteration_tikhonov_energy = print_iteration_tikhonov_energy
            self.per_iteration_flags = [print_max_warp_update, print_iteration_data_energy,
                                        print_iteration_tikhonov_energy]
            self.print_per_iteration_info = any(self.per_iteration_flags)
            self.print_per_level_info = self.print_per_iteration_info or False

    class LoggingParameters:
        def __init__(self, collect_per_level_convergence_reports=False, collect_per_level_iteration_data=False):
            self.collect_per_level_convergence_reports = collect_per_level_convergence_reports
            self.collect_per_level_iteration_data = collect_per_level_iteration_data

    def __init__(self, tikhonov_term_enabled=True, gradient_kernel_enabled=True, maximum_chunk_size=8, rate=0.1,
                 maximum_iteration_count=100, maximum_warp_update_threshold=0.001, data_term_amplifier=1.0,
                 tikhonov_strength=0.2, kernel=None, verbosity_parameters=None, visualization_parameters=None,
                 logging_parameters=None, check_interval=32, comm=None, linear_resampling=False, use_graphs=True,
                 engine_options=None):
        self.verbosity_parameters = verbosity_parameters or self.VerbosityParameters()
        self.visualization_parameters = visualization_parameters  # accepted, unused: no video writers here
        self.logging_parameters = logging_parameters or self.LoggingParameters()
        self._engine = HierarchicalEngine(
            tikhonov_term_enabled, gradient_kernel_enabled, maximum_chunk_size, rate, maximum_iteration_count,
            maximum_warp_update_threshold, data_term_amplifier, tikhonov_strength,
            None if kernel is None else np.asarray(kernel, dtype=np.float64),
            compute_energy=bool(getattr(self.verbosity_parameters, "print_iteration_data_energy", False)
                                or getattr(self.verbosity_parameters, "print_iteration_tikhonov_energy", False)),
            check_interval=check_interval,
            collect_reports=self.logging_parameters.collect_per_level_convergence_reports, comm=comm,
            collect_iteration_data=self.logging_parameters.collect_per_level_iteration_data,
            linear_resampling=linear_resampling, options=dict(dict(use_graphs=use_graphs), **(engine_options or {})))
        self.hierarchy_level = 0
        self._reports = []

    @property
    def engine(self):
        """the HierarchicalEngine behind this optimizer: its knobs (engine_options.HIERARCHICAL_DEFAULTS, also settable
        through the constructor's `engine_options=dict(...)`) and `last_call`, the report of what the last call took"""
        return self._engine

    @property
    def iteration_hook(self):
        """opt-in call-back f(level, iteration, warp_field, gradient, maximum_update_length) after every iteration --
        the place where the reference calls its visualiser (hierarchical_optimizer2d.py:242-245).  warp_field is the
        level's cumulative warp AFTER the update, gradient the (filtered) gradient it was moved by, both device tensors
        [..., D].  None (default) costs nothing; with a hook every iteration is synchronised and read back."""
        return self._engine.iteration_hook

    @iteration_hook.setter
    def iteration_hook(self, hook):
        self._engine.iteration_hook = hook

    def optimize(self, canonical_field, live_field):
        """returns the cumulative warp field, interleaved [.., D] float32: a numpy array for numpy inputs, a
        device tensor when both inputs are ROCm tensors (nothing crosses PCIe then)"""
        on_device = isinstance(canonical_field, torch.Tensor) and isinstance(live_field, torch.Tensor) \
            and canonical_field.is_cuda
        if len(canonical_field.shape) != self.DIMS or len(live_field.shape) != self.DIMS:
            raise ValueError("%s expects %d-D fields" % (type(self).__name__, self.DIMS))
        canonical = as_device_field(canonical_field)
        live = as_device_field(live_field)
        warp_planar = self._engine.optimize(canonical, live)
        self.hierarchy_level = len(self._engine.level_results)
        self._print_levels()
        self._reports = [r.report for r in self._engine.level_results if getattr(r, "report", None) is not None]
        if self._engine.comm is not None and self._engine.comm.active:  # z-slab run: return the OWNED slices only
            own = self._engine.comm.layout.owned_local()
            warp_planar = warp_planar[:, own].contiguous()
        warp = dev.interleave(warp_planar)
        return warp if on_device else warp.cpu().numpy()

    # ------------------------------------------------------------------------------------------------
    def get_per_level_iteration_counts(self):
        return [r.iteration_count for r in self._engine.level_results]

    def get_per_level_maximum_updates(self):
        return [list(r.max_updates) for r in self._engine.level_results]

    def get_per_level_convergence_reports(self):
        """one ConvergenceReport per pyramid level (run_hierarchical_optimizer3d.py:104); needs
        LoggingParameters(collect_per_level_convergence_reports=True)"""
        return list(self._reports)

    def get_per_level_iteration_data(self):
        """telemetry of the last optimize() call: one OptimizationIterationData per level (warp field after every
        iteration, data-term gradient, Tikhonov-term gradient), as numpy arrays in the interleaved API layout.
        Needs LoggingParameters(collect_per_level_iteration_data=True)  (tests/test_hierarchical_optimizer2d.py:71-102)"""
        from .telemetry import OptimizationIterationData
        out = []
        for level in self._engine.iteration_data:
            def host(t):
                return None if t is None else dev.interleave(t).cpu().numpy()
            out.append(OptimizationIterationData([host(s[0]) for s in level], [host(s[1]) for s in level],
                                                 [host(s[2]) for s in level if s[2] is not None]))
        return out

    def _print_levels(self):
        vp = self.verbosity_parameters
        if not vp.print_per_iteration_info:
            return
        for level, r in enumerate(self._engine.level_results):
            for it in range(r.iteration_count):
                line = "[ITERATION %d COMPLETED]" % it
                if vp.print_max_warp_update:
                    line += " max upd. l.: %f" % r.max_updates[it]
                if vp.print_iteration_data_energy:
                    line += " norm. data energy: %f" % (1000000.0 * r.data_energies[it] / max(r.voxel_count, 1))
                if vp.print_iteration_tikhonov_energy and self._engine.tikhonov_term_enabled:  # :204-210,:237-238
                    line += " norm. tikhonov energy: %f" % (1000000.0 * 0.5 * r.tikhonov_energies[it]
                                                            / max(r.voxel_count, 1))
                print(line)
            print("[LEVEL %d COMPLETED]" % level)


class HierarchicalOptimizer2d(_HierarchicalOptimizerBase):
    DIMS = 2
