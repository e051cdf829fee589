"""HierarchicalEngine: coarse-to-fine gradient descent on a cumulative warp field, D = 2 or 3, whole volumes and
z-slabs (nonrigid_opt/hierarchical/hierarchical_optimizer2d.py:123-246).  The drop-in classes
HierarchicalOptimizer2d / 3d are thin shells around it."""
import ctypes

import numpy as np
import torch

from . import _lib, device as dev, engine_options
from .engine_common import (_CAPTURE_LOCK, _RETIRED_GRAPHS, _combine_statistics, _conv_axis_order, _retire_graphs,
                            pyramid_level_count)
from .slab import SlabComm, SlabLayout


class LevelResult:
    def __init__(self, iteration_count, max_updates, argmax, data_energies, voxel_count=0, tikhonov_energies=()):
        self.voxel_count = voxel_count
        self.tikhonov_energies = list(tikhonov_energies)  # sum |np.gradient(previous gradient)|^2 per iteration
        self.iteration_count = iteration_count
        self.max_updates = max_updates
        self.argmax = argmax
        self.data_energies = data_energies
        self.iteration_limit_reached = False


class HierarchicalEngine:
    """coarse-to-fine gradient descent on a cumulative warp field
    (nonrigid_opt/hierarchical/hierarchical_optimizer2d.py:123-246), D = 2 or 3."""

    def __init__(self, tikhonov_term_enabled, gradient_kernel_enabled, maximum_chunk_size, rate,
                 maximum_iteration_count, maximum_warp_update_threshold, data_term_amplifier, tikhonov_strength,
                 kernel, compute_energy=False, check_interval=32, collect_reports=False, comm=None,
                 collect_iteration_data=False, linear_resampling=False, options=None):
        # use_graphs, graph_max_voxels, fused_filter, fused_filter_min_voxels, defer_maximum, blocked_levels
        engine_options.apply(self, engine_options.HIERARCHICAL_DEFAULTS, options)
        self.graph_max_voxels = int(self.graph_max_voxels)
        self.last_call = engine_options.new_call_report()
        # False: levels without a captured graph run eagerly (same results) -- set while optimizers work side by side in
        # several host threads: HIP refuses ordinary calls of OTHER threads while a capture is in progress
        self.allow_graph_capture = True
        self._graphs = {}
        self.linear_resampling = linear_resampling  # ResamplingStrategy.LINEAR (3-D): math_utils/resampling.py
        self.collect_reports = collect_reports
        self.collect_iteration_data = collect_iteration_data  # telemetry: per-iteration warp / gradient snapshots
        self.iteration_data = []
        # opt-in per-iteration call-back, f(level, iteration, warp, gradient, max_update) with device tensors in the API
        # layout [..., D]: where the reference calls its visualiser inside the loop (hierarchical_optimizer2d.py:242-245).
        # None (default): nothing is synchronised or copied per iteration; set: every iteration is read back at once.
        self.iteration_hook = None
        self.comm = comm  # SlabComm of the FINEST level (z-slab runs), or None
        self.maximum_chunk_size = maximum_chunk_size
        self.rate = rate
        self.data_term_amplifier = data_term_amplifier
        # enable-flag folding of hierarchical_optimizer2d.py:96-107
        if tikhonov_term_enabled:
            self.tikhonov_strength = tikhonov_strength
            self.tikhonov_term_enabled = tikhonov_strength != 0.0
        else:
            self.tikhonov_strength = 0.0
            self.tikhonov_term_enabled = False
        if gradient_kernel_enabled:
            self.gradient_kernel = kernel
            self.gradient_kernel_enabled = kernel is not None
        else:
            self.gradient_kernel = None
            self.gradient_kernel_enabled = False
        self.maximum_warp_update_threshold = maximum_warp_update_threshold
        self.maximum_iteration_count = int(maximum_iteration_count)
        self.compute_energy = compute_energy
        self.check_interval = max(1, int(check_interval))
        self.level_results = []
        self.last_gradient = None  # planar gradient of the finest level after the last iteration

    # ------------------------------------------------------------------------------------------------
    def _slab(self):
        return self.comm is not None and self.comm.active

    def _slab_comm_of(self, layout):
        for c in getattr(self, "_level_comms", []) or [self.comm]:
            if c is not None and c.layout is layout:
                return c
        return self.comm

    def build_pyramids(self, canonical, live):
        """canonical / live pyramids, coarsest first; live is packed with its full-resolution np.gradient
        BEFORE restriction (gradients are averaged, not recomputed: hierarchical_optimizer2d.py:126-131).
        Returns (canonical levels, packed levels, per-level SlabComm or None)."""
        if not self._slab():
            n_levels = pyramid_level_count(live.shape, self.maximum_chunk_size)
            canon_levels = [canonical]
            packed_levels = [dev.pack_live_gradient(live)]
            restrict = dev.downsample2x_linear if self.linear_resampling else dev.restrict_mean
            for _ in range(1, n_levels):
                canon_levels.append(restrict(canon_levels[-1], 1))
                packed_levels.append(restrict(packed_levels[-1], 4))
            canon_levels.reverse()
            packed_levels.reverse()
            return canon_levels, packed_levels, [None] * n_levels
        # z-slab: every level keeps `halo` neighbour slices; a level's owned slices are the restriction of the finer
        # level's owned slices (slab boundaries are multiples of 2^levels), its halos come from one exchange per level
        L0 = self.comm.layout
        if live.dim() != 3 or live.shape[0] != L0.nz_local:
            raise ValueError("slab runs need 3-D local fields with %d slices, got %r" % (L0.nz_local, tuple(live.shape)))
        global_shape = (L0.nz_global,) + tuple(live.shape[1:])
        n_levels = pyramid_level_count(global_shape, self.maximum_chunk_size)
        per = L0.z1 - L0.z0
        if per % (1 << (n_levels - 1)) != 0 or (per >> (n_levels - 1)) < max(L0.halo, 1):
            raise ValueError("a slab of %d slices cannot carry %d pyramid levels with a %d-slice halo"
                             % (per, n_levels, L0.halo))
        comms = [self.comm]
        packed = dev.pack_live_gradient(live)
        # the outermost halo slice got a one-sided z difference: refresh the halos from their owners
        comms[0].exchange_halos([packed.view(packed.shape[0], packed.shape[1], -1)])
        canon_levels, packed_levels = [canonical], [packed]
        for k in range(1, n_levels):
            fine_comm = comms[-1]
            Lf = fine_comm.layout
            Lc = SlabLayout(Lf.nz_global // 2, Lf.rank, Lf.world, Lf.halo)
            cc = SlabComm(Lc, fine_comm.group)
            own_f = Lf.owned_local()
            if self.linear_resampling:
                c_own = self._restrict_linear_owned(canon_levels[-1], Lf, 1)
                p_own = self._restrict_linear_owned(packed_levels[-1], Lf, 4)
            else:
                c_own = dev.restrict_mean(canon_levels[-1][own_f].contiguous(), 1)
                p_own = dev.restrict_mean(packed_levels[-1][own_f].contiguous(), 4)
            c_loc = torch.zeros((Lc.nz_local,) + tuple(c_own.shape[1:]), dtype=torch.float32, device=live.device)
            p_loc = torch.zeros((Lc.nz_local,) + tuple(p_own.shape[1:]), dtype=torch.float32, device=live.device)
            c_loc[Lc.owned_local()] = c_own
            p_loc[Lc.owned_local()] = p_own
            cc.exchange_halos([c_loc])
            cc.exchange_halos([p_loc.view(p_loc.shape[0], p_loc.shape[1], -1)])
            canon_levels.append(c_loc)
            packed_levels.append(p_loc)
            comms.append(cc)
        canon_levels.reverse()
        packed_levels.reverse()
        comms.reverse()
        self._level_comms = comms
        return canon_levels, packed_levels, comms

    @staticmethod
    def _restrict_linear_owned(fine, layout, channels):
        """LINEAR restriction (4x4x4 windows, math_utils/resampling.py:90-109) of a slab's owned slices: the window of a
        coarse slice reaches one fine slice past the owned range -- the neighbour's slice from the halo, or the edge
        slice again where the volume ends (the kernel's clamp).  Two slices are put on either side so that the window
        origin stays even; the outer one and the two extra coarse slices it produces are never looked at."""
        own = layout.owned_local()
        below = fine[own.start - 1:own.start] if layout.halo_lo >= 1 else fine[own.start:own.start + 1]
        above = fine[own.stop:own.stop + 1] if layout.halo_hi >= 1 else fine[own.stop - 1:own.stop]
        padded = torch.cat([below, below, fine[own], above, above], 0).contiguous()
        return dev.downsample2x_linear(padded, channels)[1:-1].contiguous()

    def optimize(self, canonical, live):
        """canonical, live: float32 device tensors [z,]y,x (z-slab runs: the local slab incl. halos).
        Returns the warp field, PLANAR [c][z][y][x] (z-slab runs: local extent, only owned slices are meaningful)."""
        if canonical.shape != live.shape:
            raise ValueError("canonical and live fields must have the same shape")
        dims = live.dim()
        self.last_call = engine_options.new_call_report()
        canon_levels, packed_levels, comms = self.build_pyramids(canonical, live)
        self.level_results = []
        self._pending_levels = []
        self.iteration_data = []
        # z-slab runs: levels whose gather operand had to be replicated on every rank because the cumulative warp
        # outgrew the halo (optimize_level); once a level needed it the finer ones start that way -- warps are not
        # rescaled between levels (hierarchical_optimizer2d.py:155-156), so they only grow
        self.replicated_levels = 0
        warp = None
        for level, (canon_l, packed_l, comm_l) in enumerate(zip(canon_levels, packed_levels, comms)):
            if level == 0:
                warp = torch.zeros((dims,) + tuple(canon_l.shape), dtype=torch.float32, device=live.device)
            self.optimize_level(canon_l, packed_l, warp, comm_l)
            if level != len(canon_levels) - 1:
                if self.linear_resampling:
                    if comm_l is not None:
                        # the lerp of a slab's first / last fine slices reads the neighbour's adjacent coarse slice;
                        # iterations never touch the warp's halo slices (the warp is only read voxel by voxel)
                        comm_l.exchange_halos([warp], width=1)
                    fine = torch.stack([dev.upsample2x_linear(warp[c].contiguous()) for c in range(dims)])
                else:
                    fine = dev.prolong_repeat(warp)
                if comm_l is not None:  # keep [owned + halo] of the finer level's layout
                    lo = comm_l.layout.halo_lo
                    fine = fine[:, lo:lo + comms[level + 1].layout.nz_local].contiguous()
                warp = fine
        self._finish_pending_levels()
        return warp

    # ------------------------------------------------------------------------------------------------
    # One level.  Gradient buffers: F[0], F[1] alternate as "previous gradient" / "this iteration's final gradient"
    # (iteration i reads F[i % 2], leaves its result in F[(i + 1) % 2]); with a gradient kernel the raw gradient and the
    # intermediate filter passes ping-pong between two scratch buffers and the LAST pass writes F[(i + 1) % 2].  The
    # buffer roles therefore repeat with period 2, which is what lets a batch of iterations be captured ONCE as a HIP
    # graph and replayed (launch-bound levels: 2-D fields, coarse 3-D levels).
    class _Level:
        pass

    def _make_level(self, canonical, packed, warp, grid, full_grid, n_records, packed_global=None):
        """packed_global: the packed live field of the WHOLE level (every rank's owned slices, SlabComm.all_gather_owned)
        for the gather instead of the local slab + halo"""
        lv = HierarchicalEngine._Level()
        dims = canonical.dim()
        tik, ker = self.tikhonov_term_enabled, self.gradient_kernel_enabled
        lv.canonical, lv.packed, lv.warp, lv.grid, lv.full_grid, lv.dims = canonical, packed, warp, grid, full_grid, dims
        lv.params = _lib.HierParams(float(self.data_term_amplifier), float(self.tikhonov_strength), float(self.rate),
                                    int(tik), int(not ker), int(self.compute_energy))
        lv.packed_global = packed_global
        if packed_global is not None:
            lv.params.packed_nz, lv.params.packed_z_global_offset = int(packed_global.shape[0]), 0
        lv.F = [torch.zeros_like(warp) for _ in range(2)] if (tik or ker) else []
        lv.S = [torch.zeros_like(warp) for _ in range(2)] if ker else []
        lv.report_g = torch.zeros_like(warp) if (self.collect_reports and not lv.F) else None
        lv.records = dev.new_records(n_records, canonical.device)
        f = dev.IterationLauncher(grid, lv.records, _lib.GATE_HIERARCHICAL, float(self.maximum_warp_update_threshold))
        n = dev.n_voxels(grid)
        lv.p_packed = f.pointer(packed, 4 * n, "packed live") if packed_global is None else \
            f.pointer(packed_global, packed_global.numel(), "packed live (whole level)")
        lv.p_canon = f.pointer(canonical, n, "canonical")
        lv.p_warp = f.pointer(warp, n * dims, "warp")
        lv.p_F = [f.pointer(t, n * dims, "gradient buffer") for t in lv.F]
        lv.p_S = [f.pointer(t, n * dims, "scratch buffer") for t in lv.S]
        lv.p_report = f.pointer(lv.report_g, n * dims, "gradient", allow_none=True)
        lv.params_ref = ctypes.byref(lv.params)
        lv.launcher = f
        # Deferred maximum (3-D levels whose filter runs in lsf_convolve_xyz, Tikhonov on): when the stop test cannot
        # fire (threshold <= 0) the maximum update length is only a log value, and the NEXT iteration's kernel reads the
        # gradient it belongs to anyway (as g_prev, for the Laplacian): that kernel writes it into the previous record
        # (lsf_hier_params::previous_max, an LSF_GATE_OPEN gate naming the record), and only the last iteration of a
        # batch keeps the separate maximum pass (44 us of 520 per 256^3 iteration).
        # (smaller levels, whose filter runs pass by pass: the last pass moves the warp, lsf_convolve_axis_update)
        lv.defer_max = (dims == 3 and tik and ker and float(self.maximum_warp_update_threshold) <= 0.0
                        and ((self.fused_filter and n >= self.fused_filter_min_voxels
                              and dev.convolve_xyz_ok(grid, self.gradient_kernel))
                             or dev.convolve_axis_update_ok(grid, self.gradient_kernel))
                        and self.defer_maximum)
        if lv.defer_max:
            lv.params_prevmax = _lib.HierParams.from_buffer_copy(lv.params)
            lv.params_prevmax.previous_max = 1
            lv.params_prevmax_ref = ctypes.byref(lv.params_prevmax)
            base = lv.records.data_ptr()
            lv.open_gates = [_lib.Gate(base + i * _lib.RECORD_BYTES, _lib.GATE_OPEN, 0.0, 0.0) for i in range(n_records)]
            lv.open_gate_refs = [ctypes.byref(g) for g in lv.open_gates]
        return lv

    def _graph_key(self, canonical):
        K = min(self.check_interval, self.maximum_iteration_count)
        return (tuple(canonical.shape), canonical.device, K - K % 2)

    def invalidate_graphs(self):
        """a setting changed: captured graphs hold the old rate / threshold / taps / iteration counts"""
        _retire_graphs(self._graphs)

    def __del__(self):
        try:
            _retire_graphs(self._graphs)
        except Exception:  # noqa: BLE001 -- interpreter shutdown: nothing left to protect
            pass

    def _enqueue(self, lv, rec_idx, prev_idx, parity, comm=None, defer_max=False, prev_deferred=False):
        """one iteration: record slot rec_idx, gated on record prev_idx (None: always runs), buffer parity 0/1.
        defer_max: leave this iteration's maximum to the next one (see _make_level); prev_deferred: the previous did"""
        f = lv.launcher
        tik, ker = self.tikhonov_term_enabled, self.gradient_kernel_enabled
        gate_ref = f.gate_ref(prev_idx)
        gate = None if prev_idx is None or prev_idx < 0 else f.gates[prev_idx]
        lib_hier = _lib.lib.lsf_hier_iteration
        if ker:
            prev, out = (lv.p_F[parity] if tik else None), lv.F[1 - parity]
            if prev_deferred and prev_idx is not None and prev_idx >= 0:
                _lib.check(lib_hier(lv.p_packed, lv.p_canon, lv.p_warp, prev, lv.p_S[0], f.grid_ref,
                                    lv.params_prevmax_ref, lv.open_gate_refs[prev_idx], f.record_ptrs[rec_idx],
                                    dev.stream_ptr()), "lsf_hier_iteration")
            else:
                _lib.check(lib_hier(lv.p_packed, lv.p_canon, lv.p_warp, prev, lv.p_S[0], f.grid_ref, lv.params_ref,
                                    gate_ref, f.record_ptrs[rec_idx], dev.stream_ptr()), "lsf_hier_iteration")
            slab = comm is not None and comm.active
            if slab:  # the z pass reads taps/2 slices of the (x,y)-filtered field on either side
                comm.exchange_halos([lv.S[0]], width=len(self.gradient_kernel) // 2)
            axes = _conv_axis_order(lv.dims)
            src = lv.S[0]
            moved = False
            if (self.fused_filter and dev.n_voxels(lv.grid) >= self.fused_filter_min_voxels
                    and dev.convolve_xyz_ok(lv.grid, self.gradient_kernel)):
                # x, y, z in one launch, which also moves the warp by its filtered gradient, component by component
                dev.convolve_xyz(src, out, lv.grid, self.gradient_kernel, gate, lv.warp, self.rate)
                axes, moved = (), True
            if len(axes) == 3 and self.fused_filter and dev.convolve_xy_ok(lv.full_grid, self.gradient_kernel):
                # smaller 3-D levels: the x and the y pass in one launch (these levels are launch-bound)
                dev.convolve_xy(src, lv.S[1], lv.full_grid, self.gradient_kernel, gate)
                src, axes = lv.S[1], [None, None, 2]
            for k, axis in enumerate(axes):
                if axis is None:
                    continue
                dst = out if k == len(axes) - 1 else lv.S[(k + 1) % 2]
                if k == len(axes) - 1 and dev.convolve_axis_update_ok(lv.grid, self.gradient_kernel):
                    # the last pass moves the warp by the gradient it writes (one launch reads and writes the warp
                    # instead of lsf_hier_update reading the gradient again)
                    dev.convolve_axis_update(src, dst, lv.warp, self.rate, lv.grid if axis == 2 else lv.full_grid, axis,
                                             self.gradient_kernel, gate)
                    moved = True
                else:
                    dev.convolve_axis(src, dst, None, lv.grid if axis == 2 else lv.full_grid, axis,
                                      self.gradient_kernel, gate)
                src = dst
            if not (moved and defer_max):
                dev.hier_update(out, None if moved else lv.warp, lv.grid, self.rate, gate, lv.records, rec_idx)
        elif tik:
            _lib.check(lib_hier(lv.p_packed, lv.p_canon, lv.p_warp, lv.p_F[parity], lv.p_F[1 - parity], f.grid_ref,
                                lv.params_ref, gate_ref, f.record_ptrs[rec_idx], dev.stream_ptr()),
                       "lsf_hier_iteration")
        else:
            _lib.check(lib_hier(lv.p_packed, lv.p_canon, lv.p_warp, None, lv.p_report, f.grid_ref, lv.params_ref,
                                gate_ref, f.record_ptrs[rec_idx], dev.stream_ptr()), "lsf_hier_iteration")

    def _final_gradient(self, lv, n_exec):
        if lv.F and n_exec:
            return lv.F[n_exec % 2]  # iteration n_exec - 1 wrote F[((n_exec - 1) + 1) % 2]
        return lv.report_g

    def _finish_level(self, lv, n_exec, dec, slab_layout=None):
        thr = float(self.maximum_warp_update_threshold)
        n_vox = dev.n_voxels(lv.grid) if slab_layout is None else slab_layout.nz_global * lv.grid.ny * lv.grid.nx
        res = LevelResult(n_exec, [float(v) for v in dec["max_value"][:n_exec]],
                          [int(v) for v in dec["argmax"][:n_exec]],
                          [float(v) for v in dec["data_energy"][:n_exec]], n_vox,
                          [float(v) for v in dec["smoothing_energy"][:n_exec]])
        res.iteration_limit_reached = n_exec >= self.maximum_iteration_count
        self.level_results.append(res)
        self.last_gradient = self._final_gradient(lv, n_exec)
        if self.collect_reports and slab_layout is None:
            # per-level ConvergenceReport (cpp get_per_level_convergence_reports, run_hierarchical_optimizer3d.py:104):
            # statistics of the last iteration's update field and of |canonical - resampled live| at this level
            from .convergence_report import (ConvergenceReport, build_tsdf_difference_statistics,
                                             build_warp_delta_statistics)
            resampled = dev.warp_field(lv.packed[..., 0].contiguous(), lv.warp, 1.0)
            g_final = self.last_gradient if self.last_gradient is not None else torch.zeros_like(lv.warp)
            res.report = ConvergenceReport(n_exec, res.iteration_limit_reached,
                                           build_warp_delta_statistics(g_final, lv.canonical, resampled, thr,
                                                                       float("inf")),
                                           build_tsdf_difference_statistics(lv.canonical, resampled))
        elif self.collect_reports:
            # z-slab: the same statistics over the OWNED slices (global voxel indices through z_global_offset), then
            # combined over the ranks -- counts and sums add, minima / maxima compare, the arg-max of the larger
            # maximum wins (smallest index on a tie, as np.argmax over the whole volume)
            from .convergence_report import (ConvergenceReport, tsdf_difference_statistics_from_raw,
                                             warp_delta_statistics_from_raw)
            L = slab_layout
            whole = dev.make_grid(lv.canonical.shape, 0, L.nz_local, L.z_global_offset)
            if lv.packed_global is None:
                resampled = dev.warp_field(lv.packed[..., 0].contiguous(), lv.warp, 1.0, whole)
            else:
                # the warp reaches past the halo: resample the replicated live field under the whole level's warp (every
                # rank the same work; reports are an opt-in) and keep the owned slices
                c_l = self._slab_comm_of(L)
                warp_g = torch.stack([c_l.all_gather_owned(lv.warp[c]) for c in range(lv.warp.shape[0])])
                whole_level = dev.warp_field(lv.packed_global[..., 0].contiguous(), warp_g, 1.0)
                resampled = torch.zeros_like(lv.canonical)
                resampled[L.owned_local()] = whole_level[L.z0:L.z1]
            g_final = self.last_gradient if self.last_gradient is not None else torch.zeros_like(lv.warp)
            raw = torch.stack([dev.warp_statistics(g_final, lv.canonical, resampled, thr, lv.grid),
                               dev.tsdf_difference_statistics(lv.canonical, resampled, lv.grid)])
            rows = self._slab_comm_of(L).gather_rows(raw)
            shape = (L.nz_global,) + tuple(lv.canonical.shape[1:])
            res.report = ConvergenceReport(
                n_exec, res.iteration_limit_reached,
                warp_delta_statistics_from_raw(_combine_statistics([r[0] for r in rows], has_min=False), shape, thr,
                                               float("inf")),
                tsdf_difference_statistics_from_raw(_combine_statistics([r[1] for r in rows], has_min=True), shape))

    OPEN_RECORD = 0x7F800000FFFFFFFF  # packed max = +inf: "previous iteration has not converged" for the gate

    def optimize_level(self, canonical, packed, warp, comm=None):
        slab = comm is not None and comm.active
        max_it = self.maximum_iteration_count
        n_vox = canonical.numel()
        hooked = self.iteration_hook is not None
        if (self.blocked_levels and canonical.dim() == 2 and not slab and not hooked and not self.collect_iteration_data
                and not self.compute_energy and max_it >= 1
                and (not self.gradient_kernel_enabled or len(self.gradient_kernel) in dev.XYZ_TAP_COUNTS)):
            return self._optimize_level_blocked(canonical, packed, warp)
        if (self.use_graphs and not slab and not self.collect_iteration_data and not hooked and max_it >= 4
                and self.check_interval >= 2 and n_vox <= self.graph_max_voxels
                and (self.allow_graph_capture or self._graph_key(canonical) in self._graphs)):
            return self._optimize_level_graph(canonical, packed, warp)
        if slab:
            L = comm.layout
            grid = dev.make_grid(canonical.shape, L.z_begin, L.z_end, L.z_global_offset)
            full_grid = dev.make_grid(canonical.shape, 0, L.nz_local, L.z_global_offset)
            reach = len(self.gradient_kernel) // 2 if self.gradient_kernel_enabled else 0
            if L.halo < max(reach, 2):
                raise ValueError("slab halo of %d slices is too narrow: this configuration needs >= %d"
                                 % (L.halo, max(reach, 2)))
        else:
            L = None
            grid = full_grid = dev.make_grid(canonical.shape)
        thr = float(self.maximum_warp_update_threshold)
        tik = self.tikhonov_term_enabled
        packed_global = warp_at_start = None
        if slab and getattr(self, "replicated_levels", 0) > 0:
            packed_global = comm.all_gather_owned(packed)
            self.replicated_levels += 1
        elif slab:
            warp_at_start = warp.clone()  # what a restart of this level on the replicated field begins from
        lv = self._make_level(canonical, packed, warp, grid, full_grid, max(max_it, 1), packed_global)
        records = lv.records
        snapshots = []
        it = 0
        n_exec = 0
        dec = None
        while it < max_it:
            batch = 1 if hooked else min(self.check_interval, max_it - it)
            for i in range(it, it + batch):
                if self.collect_iteration_data:
                    # telemetry (cpp LoggingParameters.collect_per_level_iteration_data): the two gradient terms
                    # the reference hands to its visualiser (hierarchical_optimizer2d.py:196,202,242-245) are
                    # produced by two extra launches of the same kernel on the pre-update warp
                    gate = lv.launcher.gates[i - 1] if i > 0 else None
                    d_snap = torch.zeros_like(warp)
                    gathered = packed if packed_global is None else packed_global
                    wide = (0, 0, 0, 0) if packed_global is None else (0, 0, int(packed_global.shape[0]), 0)
                    dev.hier_iteration(gathered, canonical, warp, None, d_snap, grid,
                                       _lib.HierParams(1.0, 0.0, 0.0, 0, 0, 0, *wide), gate, records, i)
                    t_snap = None
                    if tik:
                        t_snap = torch.zeros_like(warp)  # = laplace(previous gradient): 0*gd - (-1)*lap
                        dev.hier_iteration(gathered, canonical, warp, lv.F[i % 2], t_snap, grid,
                                           _lib.HierParams(0.0, -1.0, 0.0, 1, 0, 0, *wide), gate, records, i)
                    snapshots.append([None, d_snap, t_snap])
                defer = lv.defer_max and not slab and not hooked and not self.collect_iteration_data
                self._enqueue(lv, i, i - 1 if i > 0 else None, i % 2, comm, defer_max=defer and i + 1 < it + batch,
                              prev_deferred=defer and i > it)
                if self.collect_iteration_data:
                    snapshots[-1][0] = warp.clone()
                if slab:
                    if tik:  # the next iteration's Laplacian reads one slice of this gradient on either side
                        comm.exchange_halos([lv.F[(i + 1) % 2]], width=1)
                    if i + 1 < max_it:
                        comm.reduce_max(records, i)  # the next iteration's gate tests the GLOBAL max
            if slab:
                comm.reduce_records(records, it, it + batch)
            it += batch
            dec = dev.decode_records(dev.records_to_host(records[:it]))  # the only host sync of the batch
            n_exec = int(dec["executed"].sum())
            if slab and packed_global is None:
                # the gather follows the cumulative warp: it must stay inside the halo of the static packed field.  When it
                # does not, the reference does not stop either (hierarchical_optimizer2d.py:169-171 tests the update
                # threshold only): every rank sees the same reduced maximum, so all of them together discard this level's
                # iterations, replicate the level's packed field (5 x 512 MiB at 512^3 against 288 GB) and run
                # the level again from the warp it started with -- the gather then never leaves the device
                wz = warp[2][L.owned_local()].abs().max().reshape(1)
                comm.reduce_scalar_max(wz)
                if not (float(wz.item()) < L.halo - 1):
                    warp.copy_(warp_at_start)
                    self.replicated_levels = 1
                    return self.optimize_level(canonical, packed, warp, comm)
            if hooked and n_exec == it:  # iteration it - 1 ran: its gradient is in the buffer the next one reads
                g_now = lv.F[it % 2] if lv.F else lv.report_g
                own = (slice(None), L.owned_local()) if slab else (slice(None),)
                self.iteration_hook(len(self.level_results), it - 1, dev.interleave(warp[own].contiguous()),
                                    dev.interleave(g_now[own].contiguous()), float(dec["max_value"][it - 1]))
            if n_exec < it or dec["max_value"][n_exec - 1] < np.float32(thr):
                break
        if dec is None:  # maximum_iteration_count == 0: the reference's loop body never runs
            dec = dev.decode_records(dev.records_to_host(records[:1]))
        if self.collect_iteration_data:
            self.iteration_data.append(snapshots[:n_exec])  # snapshots of gated (not executed) launches are dropped
        self._finish_level(lv, n_exec, dec, L)
        return warp

    # ------------------------------------------------------------------------------------------------
    BLOCKED_ITERATIONS_PER_LAUNCH = 8

    def _optimize_level_blocked(self, canonical, packed, warp):
        """2-D levels, no energy printouts: the whole level in ONE foreign call, K = 8 iterations
        per launch advanced inside LDS tile by tile (lsf_hier_level_run_2d: temporal blocking -- a 512^2 level is
        launch-bound, 7.5 us per iteration from a HIP graph against ~1 us of work).  With the gradient kernel (the
        reference's default constructor) the launch also runs the filter's two passes and the update behind them -- one
        launch instead of four per iteration -- and an iteration consumes taps / 2 + 1 rings of a tile's surroundings: K = 2
        for seven taps.  Same arithmetic on the same inputs:
        warp, gradient and every iteration's maximum equal the per-iteration path's (tests/test_gpu_blocked_levels.py).
        A stop test that can fire (threshold > 0, hierarchical_optimizer2d.py:169-171) is looked at launch by launch on
        the card; the launch in which the level converged is then repeated from its untouched inputs with the reference's
        number of iterations, so the level ends exactly where the reference's ends."""
        max_it = self.maximum_iteration_count
        grid = dev.make_grid(canonical.shape)
        n = dev.n_voxels(grid)
        K = self.BLOCKED_ITERATIONS_PER_LAUNCH
        taps, n_taps = None, 0
        if self.gradient_kernel_enabled:
            kernel = np.ascontiguousarray(np.asarray(self.gradient_kernel, dtype=np.float64))
            n_taps = int(kernel.size)
            if min(canonical.shape) < n_taps:  # (the reference cannot do this either: np.convolve's 'same' mode)
                raise ValueError("cannot convolve a field of extent %d with a %d-tap kernel" % (min(canonical.shape), n_taps))
            taps = kernel.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        # rings of a tile's surroundings an iteration consumes: one for the Tikhonov term's Laplacian, taps / 2 for the filter
        rings = int(bool(self.tikhonov_term_enabled)) + n_taps // 2
        if rings:
            K = max(1, K // rings)
        thr = np.float32(self.maximum_warp_update_threshold)
        gated = bool(thr > 0.0)
        warps = [warp, torch.empty_like(warp)]
        F = [torch.zeros_like(warp), torch.empty_like(warp)]
        records = dev.new_records(max_it, canonical.device)
        params = _lib.HierParams(float(self.data_term_amplifier), float(self.tikhonov_strength), float(self.rate),
                                 int(bool(self.tikhonov_term_enabled)), int(n_taps == 0), 0)
        p_packed, p_canonical = dev._ptr(packed, 4 * n, "packed live"), dev._ptr(canonical, n, "canonical")
        p_warp = [dev._ptr(w, 2 * n, "warp") for w in warps]
        p_g = [dev._ptr(g, 2 * n, "gradient") for g in F]

        def run(first, record_ptr, iterations, threshold):  # launches reading pair `first` first
            _lib.check(_lib.lib.lsf_hier_level_run_2d(
                p_packed, p_canonical, p_warp[first], p_warp[1 - first], p_g[first], p_g[1 - first], ctypes.byref(grid),
                ctypes.byref(params), taps, n_taps, record_ptr, iterations, K, threshold, dev.stream_ptr()),
                "lsf_hier_level_run_2d")

        run(0, ctypes.c_void_p(records.data_ptr()), max_it, float(thr) if gated else 0.0)
        launches = (max_it + K - 1) // K
        dec = None
        if gated:
            dec = dev.decode_records(dev.records_to_host(records[:max_it]))
            ran = int(dec["executed"].sum())  # whole launches: a multiple of K, or every iteration
            below = np.nonzero(dec["max_value"][:ran] < thr)[0]
            n_exec = int(below[0]) + 1 if below.size else ran
            last = (n_exec - 1) // K  # the launch the level ended in; the ones behind it were no-ops
            if n_exec < min((last + 1) * K, max_it):
                # ... in the middle of it: once more from its inputs, as many iterations as the reference runs
                run(last % 2, ctypes.c_void_p(dev.new_records(K, canonical.device).data_ptr()), n_exec - last * K, 0.0)
            launches = last + 1
        if launches % 2:
            warp.copy_(warps[1])
        lv = HierarchicalEngine._Level()
        lv.canonical, lv.packed, lv.warp, lv.grid, lv.full_grid, lv.dims = canonical, packed, warp, grid, grid, 2
        final = F[launches % 2]
        lv.F = [final, final]  # _final_gradient reads F[n_exec % 2]
        lv.report_g = None
        self.last_call.blocked_levels += 1
        if gated:
            # (the level's results are made of `dec` at the end of optimize(): the card has nothing queued right now)
            self._pending_levels.append((lv, dec, n_exec))
            self.level_results.append(None)
            return warp
        # nothing on the host depends on this level's records (a fixed count): they are read with the other levels' at the
        # end of optimize() -- ONE synchronisation per call instead of one per level, with the next level's launches
        # already queued behind this one's
        self._pending_levels.append((lv, records, max_it))
        self.level_results.append(None)
        return warp

    def _finish_pending_levels(self):
        """the records of the levels _optimize_level_blocked left unread: one transfer, then every level's results"""
        pending, self._pending_levels = self._pending_levels, []
        if not pending:
            return
        if isinstance(pending[0][1], dict):  # levels with a stop test: their records were read level by level
            first = self.level_results.index(None)
            self.level_results = self.level_results[:first]
            for lv, dec, n_exec in pending:
                self._finish_level(lv, n_exec, dec)
            return
        used = torch.cat([dev.slot_view(records)[:n, :, :dev.USED_SLOT_WORDS].reshape(-1) for _, records, n in pending])
        host = dev.pinned_scratch("level records", used.numel(), torch.int64)[:used.numel()]
        host.copy_(used, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        flat, at = host.numpy(), 0
        slots = dev.slot_view(pending[0][1]).shape[1]
        first = self.level_results.index(None)
        results, self.level_results = self.level_results, self.level_results[:first]
        for lv, records, n in pending:
            words = n * slots * dev.USED_SLOT_WORDS
            dec = dev.decode_records(flat[at:at + words].reshape(n, slots, dev.USED_SLOT_WORDS).copy())
            at += words
            self._finish_level(lv, int(dec["executed"].sum()), dec)
        assert len(self.level_results) == len(results)

    def _optimize_level_graph(self, canonical, packed, warp):
        """launch-bound levels: K iterations (K even) are captured once per level shape as a HIP graph over persistent
        buffers and replayed; a replay costs one launch instead of K x (1..5).  Record slots 0..K-1 form a ring that the
        graph itself re-zeroes, slot K keeps the previous batch's last record for the first gate of the next batch, so
        iteration counts and results are exactly those of the eager path (tests demand equality)."""
        max_it = self.maximum_iteration_count
        thr = np.float32(self.maximum_warp_update_threshold)
        key = self._graph_key(canonical)
        K = key[2]
        entry = self._graphs.get(key)
        if entry is None:
            grid = dev.make_grid(canonical.shape)
            lv = self._make_level(torch.empty_like(canonical), torch.empty_like(packed), torch.empty_like(warp), grid,
                                  grid, K + 1)
            side = torch.cuda.Stream(device=canonical.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up launch outside capture (first-use initialisation of the kernels)
                lv.warp.zero_()
                lv.canonical.zero_()
                lv.packed.zero_()
                self._enqueue(lv, 0, None, 0)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread_local: another optimizer working in another host thread on another stream (experiment/multipair.py:
            # pairs in flight) must not invalidate this capture; two captures at once are kept apart by the lock
            with _CAPTURE_LOCK:
                del _RETIRED_GRAPHS[:]  # graphs of engines that are gone die here, with no capture in progress
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    lv.records[K].copy_(lv.records[K - 1])
                    lv.records[:K].zero_()
                    for j in range(K):
                        # (a stop test that cannot fire: an iteration's maximum is left to the next one's first kernel,
                        # the batch's last keeps its own pass -- see _make_level)
                        self._enqueue(lv, j, K if j == 0 else j - 1, j % 2, defer_max=lv.defer_max and j + 1 < K,
                                      prev_deferred=lv.defer_max and j > 0)
            entry = self._graphs[key] = (lv, graph)
        lv, graph = entry
        lv.canonical.copy_(canonical)
        lv.packed.copy_(packed)
        lv.warp.copy_(warp)
        for t in lv.F + lv.S:
            t.zero_()
        lv.records.zero_()
        dev.set_record_max(lv.records, K - 1, HierarchicalEngine.OPEN_RECORD)
        done, n_exec, converged = 0, 0, False
        parts = []
        while done + K <= max_it and not converged:
            graph.replay()
            dec = dev.decode_records(dev.records_to_host(lv.records[:K]))  # host sync once per K iterations
            k_exec = int(dec["executed"].sum())
            parts.append({k: v[:k_exec].copy() for k, v in dec.items()})
            n_exec += k_exec
            done += K
            converged = k_exec < K or dec["max_value"][k_exec - 1] < thr
        rest = max_it - done
        if not converged and rest > 0:  # the remainder of a limit that is not a multiple of K: eager, same buffers
            rem = self._make_level(lv.canonical, lv.packed, lv.warp, lv.grid, lv.full_grid, rest + 1)
            rem.F, rem.S, rem.report_g = lv.F, lv.S, lv.report_g
            rem.p_F, rem.p_S, rem.p_report = lv.p_F, lv.p_S, lv.p_report
            rem.records[0].copy_(lv.records[K - 1])
            for t in range(rest):
                self._enqueue(rem, t + 1, t, (done + t) % 2)
            dec = dev.decode_records(dev.records_to_host(rem.records[1:rest + 1]))
            k_exec = int(dec["executed"].sum())
            parts.append({k: v[:k_exec].copy() for k, v in dec.items()})
            n_exec += k_exec
        merged = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]} if parts else \
            dev.decode_records(np.zeros((1, dev.RECORD_WORDS), np.int64))
        warp.copy_(lv.warp)
        final_level = lv
        self._finish_level(final_level, n_exec, merged)
        # _finish_level looked at the persistent buffers; hand out copies so that the next optimize() cannot alias them
        if self.last_gradient is not None:
            self.last_gradient = self.last_gradient.clone()
        return warp
