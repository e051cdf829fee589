"""Device-resident optimizer engines (dimension-generic, slab-aware).  The drop-in classes under
nonrigid_opt/ are thin shells around these.

All state lives in torch ROCm tensors; every per-voxel operation is a hand-written HIP kernel reached through
the C ABI (device.py -> liblsf_hip.so).  The host loop only enqueues launches and, every `check_interval`
iterations, reads back the tiny iteration records to learn whether the device-side convergence gate has closed.
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib, device as dev
from .slab import SlabComm, SlabLayout


def as_device_field(x, device=None):
    """numpy array or torch tensor -> contiguous float32 ROCm tensor (no copy when already one)"""
    dev.require_gpu()
    if isinstance(x, torch.Tensor):
        t = x
        if not t.is_cuda:
            t = t.to(device or "cuda")
        return t.to(torch.float32).contiguous()
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    return torch.from_numpy(a).to(device or "cuda")


def _is_power_of_two(n):
    return n > 0 and (n & (n - 1)) == 0


def pyramid_level_count(shape, maximum_chunk_size):
    """level-count rule and error behaviour of nonrigid_opt/hierarchical/pyramid.py:31-45"""
    if not all(_is_power_of_two(int(s)) for s in shape):
        raise ValueError("The argument 'field' must be an array where each dimension is a power of two.")
    if not _is_power_of_two(int(maximum_chunk_size)):
        raise ValueError("The argument 'maximum_chunk_size' must be an integer power of 2, i.e. 4, 8, 16, etc.")
    p = int(maximum_chunk_size).bit_length() - 1
    if min(int(s).bit_length() - 1 for s in shape) <= p:
        raise ValueError("maximum chunk size {:d} is too large for a field of size {:s}"
                         .format(int(maximum_chunk_size), str(tuple(int(s) for s in shape))))
    return p + 1


def _conv_axis_order(dims):
    # kernel axis ids: 0 = x, 1 = y, 2 = z.  2-D: y then x (math_utils/convolution.py:77-83);
    # 3-D: x, y, z (math_utils/convolution.py:94-105)
    return [1, 0] if dims == 2 else [0, 1, 2]



# HIP graphs and host threads (experiment/multipair.py runs optimizers side by side, a thread and a stream each): captures
# are thread-local and one at a time (the lock); and a captured graph is never DESTROYED while another thread captures --
# torch's graph destructor synchronises the device, which a capture in progress turns into a fatal error, and Python may
# finalise an abandoned optimizer in any thread at any time.  Engines therefore retire their graphs into a list that is
# emptied under the lock, right before the next capture (or never: a few KB each).
_CAPTURE_LOCK = threading.Lock()
_RETIRED_GRAPHS = []


def _retire_graphs(graphs):
    _RETIRED_GRAPHS.extend(graphs.values())  # list.extend is atomic under the GIL
    graphs.clear()

class _Counted:
    """stands for a band list where only the number of listed voxels matters"""

    def __init__(self, count):
        self.count = int(count)


class _Lazy:
    """a value made on first use"""

    def __init__(self, make):
        self._make, self._value = make, None

    def get(self):
        if self._value is None:
            self._value = self._make()
            self._make = None
        return self._value


def _combine_statistics(rows, has_min):
    """per-rank raw statistics (the 8 doubles of lsf_warp_statistics / lsf_tsdf_difference_statistics over disjoint
    z-ranges, arg-max as GLOBAL voxel index) -> the statistics of the union"""
    rows = [np.asarray(r, dtype=np.float64) for r in rows]
    out = np.zeros(8)
    sums = (0, 3, 4) if has_min else (0, 1, 3, 4)
    for k in sums:
        out[k] = sum(r[k] for r in rows)
    if has_min:
        out[1] = min(r[1] for r in rows)
    best = max(rows, key=lambda r: (r[2], -r[5] if r[5] >= 0 else -np.inf))
    out[2], out[5] = best[2], best[5]
    return out


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
                 collect_iteration_data=False, linear_resampling=False, use_graphs=True, graph_max_voxels=1 << 21):
        self.use_graphs = use_graphs                # HIP-graph replay for launch-bound levels
        # False: levels without a captured graph run eagerly (same results) -- set while optimizers work side by side in
        # several host threads: HIP refuses ordinary calls of OTHER threads while a capture is in progress
        self.allow_graph_capture = True
        self.graph_max_voxels = int(graph_max_voxels)  # ... i.e. levels of at most this many voxels
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
        # 3-D levels from 2^23 voxels up: lsf_convolve_xyz instead of three passes (0.21 against 0.25 ms at 256^3, 1.29
        # against 1.9 ms at 512^3; below that its 64 x 16-column blocks are too few to fill the GPU: 0.045 / 0.035 ms at 128^3)
        self.fused_filter = True  # (False: three convolve_axis passes -- measurements, tests)
        self.fused_filter_min_voxels = 1 << 23
        self.defer_maximum = True  # (False: every iteration keeps its own maximum pass -- tests hold the two against each other)
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
        canon_levels, packed_levels, comms = self.build_pyramids(canonical, live)
        self.level_results = []
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
        lv.defer_max = (dims == 3 and tik and ker and float(self.maximum_warp_update_threshold) <= 0.0
                        and self.fused_filter and n >= self.fused_filter_min_voxels
                        and dev.convolve_xyz_ok(grid, self.gradient_kernel)
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
            for k, axis in enumerate(axes):
                dst = out if k == len(axes) - 1 else lv.S[(k + 1) % 2]
                dev.convolve_axis(src, dst, None, lv.grid if axis == 2 else lv.full_grid, axis, self.gradient_kernel,
                                  gate)
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
                        self._enqueue(lv, j, K if j == 0 else j - 1, j % 2)
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


class SlavchevaOutcome:
    """final fields of one SlavchevaEngine.optimize() call, left on the device in the layout the iteration kernels use
    (the float4 state of the fused path, or planar live / warp of the Sobolev path) and handed out on demand"""

    def __init__(self, grid, canonical, state=None, live=None, warp_planar=None, listed=None, sparse=None,
                 warp_zeroed=None):
        self.grid, self.canonical, self.state = grid, canonical, state
        self._live, self._warp_planar = live, warp_planar
        # (input live field, band lists, state_prepare's unlisted counts): finalize then visits the band voxels only
        self._listed = listed
        # dev.StatePrepare whose states were initialised near the band only: readers of the WHOLE state complete it first
        self._sparse = sparse
        self._guard = None
        self._warp_zeroed = warp_zeroed  # a zero-filled API-layout warp tensor made while the card was idle (or None)

    def guard(self, records, count, limit):
        """a sparse run: the listed finalize pass leaves the caller's fields alone when one of records[0..count) holds a
        maximum update of `limit` voxels or more (it looks at the records itself, on the device)"""
        self._guard = (records, int(count), float(limit))

    def _whole_state(self):
        if self._sparse is not None:
            self._sparse.complete(self.state, self._listed[0])
            self._sparse = None
        return self.state

    def _shape(self):
        g = self.grid
        return (g.nz, g.ny, g.nx) if g.dims == 3 else (g.ny, g.nx)

    def _device(self):
        return (self.state if self.state is not None else self._live).device

    def live(self):
        if self._live is None:
            self._live = torch.empty(self._shape(), dtype=torch.float32, device=self._device())
            dev.state_unpack(self._whole_state(), self.grid, self._live, None, None)
        return self._live

    def warp_planar(self):
        if self._warp_planar is None:
            self._warp_planar = torch.empty((self.grid.dims,) + self._shape(), dtype=torch.float32,
                                            device=self._device())
            dev.state_unpack(self._whole_state(), self.grid, None, self._warp_planar, None)
        return self._warp_planar

    def finalize(self, live_out=None, lower_threshold=0.0, statistics=False):
        """ONE pass for the end of optimize(): writes the final live field into `live_out` (a contiguous float32
        device tensor, or None for a new one), builds the interleaved warp [z,]y,x,c and -- with `statistics` -- the raw
        convergence statistics (float64 [16] on the HOST: warp [0:8], |canonical - live| [8:16]).
        Returns (live, warp_interleaved, raw statistics or None)."""
        early = getattr(self, "_early", None)
        if early is not None and early[0][0] is live_out and \
                early[0][1:] == (float(lower_threshold), bool(statistics)):
            # already enqueued by optimize() behind the last iteration (fixed iteration counts): nothing left to launch
            _, target, warp, raw = early
            self._early = None
            if live_out is not None and target is not live_out:
                live_out.copy_(target)
            if raw is not None:
                host, done = raw
                done.synchronize()
                raw = host.numpy().copy()  # the pinned buffer is reused by the next call
            return target, warp, raw
        return self._finalize_now(live_out, lower_threshold, statistics, to_host=True)

    def enqueue_finalize(self, live_out, lower_threshold, statistics):
        """launch the finalize pass now (no host synchronisation); finalize() with the same arguments collects it"""
        target, warp, raw = self._finalize_now(live_out, lower_threshold, statistics, to_host=False)
        if raw is not None:  # on its way to the host behind the pass: the caller's next synchronising read covers it
            host = dev.pinned_scratch("finalize statistics", raw.numel(), raw.dtype)
            host.copy_(raw, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            raw = (host, done)
        self._early = ((live_out, float(lower_threshold), bool(statistics)), target, warp, raw)

    def _finalize_now(self, live_out, lower_threshold, statistics, to_host):
        g = self.grid
        full = dev.full_range(g)
        if live_out is None or not (live_out.is_cuda and live_out.dtype == torch.float32 and live_out.is_contiguous()):
            target = torch.empty(self._shape(), dtype=torch.float32, device=self._device())
        else:
            target = live_out
        if self.state is not None and self._listed is not None and not (statistics and self._listed[2] is None):
            # outside the band lists nothing ever moves: the input live field and a zero warp are already final there
            live0, bands, unlisted = self._listed
            unlisted = unlisted or (0, -1)
            if target is not live0:
                target.copy_(live0)
            warp, self._warp_zeroed = self._warp_zeroed, None
            if warp is None:
                warp = torch.zeros(self._shape() + (g.dims,), dtype=torch.float32, device=self._device())
            raw = dev.state_finalize_listed(self.state, self.canonical, full, bands, unlisted, target, warp,
                                            lower_threshold, statistics, guard=self._guard)
            self._live = target
        elif self.state is not None:
            warp = torch.empty(self._shape() + (g.dims,), dtype=torch.float32, device=self._device())
            raw = dev.state_finalize(self._whole_state(), self.canonical, full, target, None, warp, lower_threshold,
                                     statistics)
            self._live = target
        else:
            # planar final fields (SobolevFusion path): one pass as well (lsf_planar_finalize)
            warp = torch.empty(self._shape() + (g.dims,), dtype=torch.float32, device=self._device())
            raw = dev.planar_finalize(self._live, self._warp_planar, self.canonical, full,
                                      None if target is self._live else target, warp, lower_threshold, statistics)
        if not to_host:
            return target, warp, raw
        if live_out is not None and target is not live_out:
            live_out.copy_(target)
        return target, warp, (raw.cpu().numpy() if raw is not None else None)


class _RunLog(dict):
    """the per-iteration log of a call as the lists the callers read -- max_warps, max_warp_indices, data_energies,
    smoothing_energies, level_set_energies --, converted from the decoded records when a key is first read: the lists of a
    50-iteration call cost ~10 us of host time behind the call's last synchronisation, where the card waits for the next
    call's first launch"""

    def __init__(self, max_value, argmax, energies, weights):
        super().__init__()
        self._pending = {"max_warps": lambda: max_value.tolist(), "max_warp_indices": lambda: argmax.tolist(),
                         "data_energies": lambda: (weights[0] * energies[:, 0]).tolist(),
                         "smoothing_energies": lambda: (weights[1] * energies[:, 1]).tolist(),
                         "level_set_energies": lambda: (weights[2] * energies[:, 2]).tolist()}

    def __missing__(self, key):
        value = self[key] = self._pending.pop(key)()
        return value

    def _all(self):
        for key in list(self._pending):
            self[key]
        return self

    def keys(self):
        return dict.keys(self._all())

    def items(self):
        return dict.items(self._all())

    def values(self):
        return dict.values(self._all())

    def __iter__(self):
        return dict.__iter__(self._all())

    def __len__(self):
        return dict.__len__(self._all())

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._pending

    def __eq__(self, other):
        return dict.__eq__(self._all(), other._all() if isinstance(other, _RunLog) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class _RunOutcome(SlavchevaOutcome):
    """the final fields of a call the LIBRARY enqueued in one piece (SlavchevaEngine._optimize_run): the live field is
    already in the caller's array and the statistics are on the host; the dense API-layout warp is only built when somebody
    asks for it (the reference's optimize() does not hand its warp field out at all, slavcheva_optimizer2d.py:332-408) --
    from the final state's LISTED voxels, so that states initialised near the band only need no completion and nothing here
    depends on the caller's tensor staying as the call left it"""

    def __init__(self, grid, canonical, state, target, bands, raw):
        super().__init__(grid, canonical, state=state)
        self._live, self._bands, self._raw, self._warp = target, bands, raw, None

    def live(self):
        return self._live

    def warp_interleaved(self):
        if self._warp is None:
            warp = torch.zeros(self._shape() + (self.grid.dims,), dtype=torch.float32, device=self._device())
            dev.state_finalize_listed(self.state, self.canonical, dev.full_range(self.grid), self._bands, (0, -1), None,
                                      warp, 0.0, False)
            self._warp = warp
        return self._warp

    def warp_planar(self):
        if self._warp_planar is None:
            self._warp_planar = dev.deinterleave(self.warp_interleaved(), self.grid.dims)
        return self._warp_planar

    def finalize(self, live_out=None, lower_threshold=0.0, statistics=False):
        """(live, a callable that builds the interleaved warp, raw statistics): everything was produced by the call"""
        if live_out is not None and live_out is not self._live:
            live_out.copy_(self._live)
        return self._live, self.warp_interleaved, (self._raw if statistics else None)


class _HaloTooNarrow(Exception):
    """a z-slab run met a warp update its halo schedule cannot carry (SlavchevaEngine.optimize re-runs it wider)"""

    def __init__(self, max_update, validity):
        super().__init__("warp update of %.3f voxels against %d slice(s) of validity" % (max_update, validity))
        self.max_update = float(max_update)


class _SobolevStatePlan:
    """launch arguments of the SobolevFusion iteration on the float4 layouts (lsf_sobolev_state.hip), materialised once per
    optimize() call: iteration i reads states[i % 2] and writes the other; g4 = [raw gradient, filter buffer A, filter
    buffer B] (float4, zero-initialised: unlisted voxels are never written).  3-D: gradient -> raw, x pass raw -> A,
    y pass A -> B, z pass + update + re-warp B -> final gradient in A; 2-D: y pass raw -> A, x pass + update A -> B.
    3-D whole volumes with `boxes` (the band's LSF_BAND_ALL boxes): gradient + x pass -> A, then y pass, z pass, update and
    re-warp in ONE launch box by box (lsf_sobolev_state_update_boxes), final gradient in B."""

    STRIPS = 8  # row bands of the strip-major list the z pass walks (8 / 16 / 32 measured: profiles/r04_probe_sobolev_sweep.txt)

    def __init__(self, launcher, states, canonical, grid, params, bands, g4, taps, min_iterations, iterations_hint=0,
                 gradient_every_iteration=True, boxes=None):
        f = self.f = launcher
        n = dev.n_voxels(grid)
        self.p_state = [f.pointer(t, 4 * n, "state") for t in states]
        self.p_canon = f.pointer(canonical, n, "canonical")
        self.p_g = [f.pointer(t, 4 * n, "gradient buffer") for t in g4]
        self.g4 = g4
        self.bands = [b for b in bands if b.count] or bands[:1]
        self.params_ref = ctypes.byref(params)
        self.taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float64))
        self.p_taps = self.taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self.n_taps = int(self.taps.size)
        self.min_iterations = min_iterations
        # the filtered gradient is an OUTPUT of the last executed iteration only (gradient_field, the reference's attribute
        # of that name): a run whose iteration count is fixed stores just that one (16 B per voxel and iteration less)
        self.last_iteration = None if gradient_every_iteration else iterations_hint - 1
        self.stream = dev.stream_ptr()
        self.axes = _conv_axis_order(grid.dims)
        # 3-D: the gradient and the x pass in ONE launch (lsf_sobolev_state_gradient_x: the raw gradient never reaches
        # memory), over ONE list of the whole band -- a tap at a voxel of another list would count as zero
        self.fused_x = self.fuses_x(grid) and len(g4) == 2
        self.bands_first = self.bands
        if self.fused_x and len(self.bands) == 2:
            lo, hi = self.bands
            merged = torch.empty(lo.count + hi.count, dtype=torch.int32, device=states[0].device)
            vp, i64 = ctypes.c_void_p * 1, ctypes.c_int64 * 1
            _lib.check(_lib.lib.lsf_merge_sorted_runs(vp(lo.pointer.value), i64(lo.count), vp(hi.pointer.value),
                                                      i64(hi.count), vp(merged.data_ptr()), 1, self.stream),
                       "lsf_merge_sorted_runs")
            self.bands_first = [dev.BandList(merged, merged.numel(), _lib.BAND_ALL)]
        # index of the buffer that holds the final gradient
        self.final = (0 if self.fused_x else 1) if grid.dims == 3 else 2
        self.boxes = boxes if self.fused_x else None  # (tensor [n, 2] int64, n)
        if self.boxes is not None:
            self.final = 1
            # (the boxes regrouped strip by strip, as the z pass's list is, measured 73.4 against 72.7 us per 256^3 iteration:
            # the box kernel waits for requests, not for the fabric -- profiles/r05_sobolev_box_probes.txt)
            self.p_boxes = ctypes.c_void_p(self.boxes[0].data_ptr())
            self.bands_last = self.bands
            return
        # The LAST pass runs along z: its seven taps lie in seven slices.  In list order an XCD sweeps a z-range with a
        # window of ~2 slices of the band in flight, its 4 MB L2 cannot keep seven slices of five streams, and every tap
        # comes through the fabric (237 MB per launch at 256^3 against 130 MB of compulsory traffic: the kernel ran at the
        # Infinity Cache's 5.5 TB/s, profiles/r04_sobolev_pmc_hbm_traffic.csv).  In STRIP-major order -- eight strips of rows,
        # each swept through z -- an XCD's window spans ~19 slices of ITS strip, so the z -/+ 3 taps are lines its own CUs
        # have just read.  Results do not depend on the order (every listed voxel is written by its index); one sort per
        # call, worth it from a handful of iterations on.
        self.bands_last = self.bands
        strips = self.STRIPS
        if grid.dims == 3 and iterations_hint >= 8 and strips > 0:
            self.bands_last = [self._strip_major(b, grid, strips) if b.count >= (1 << 17) else b for b in self.bands]

    @staticmethod
    def fuses_x(grid):
        """does the iteration take the fused gradient + x pass (then two gradient buffers suffice instead of three)?"""
        return grid.dims == 3

    @staticmethod
    def _strip_major(band, grid, strips=8):
        """the ascending list regrouped strip by strip (strips of ceil(ny / strips) rows, each swept through z), ascending
        inside a strip: lsf_band_list_strip_major.  No sort: in the ascending list the entries of one (slice, strip) are ONE
        run, so run boundaries, a scan of their lengths in strip-major order and a gather do it -- three small launches in
        one host call (a radix sort of the 1.6 M keys of a 256^3 sphere pair took 0.43 ms of a 4.5 ms call with 64-bit keys,
        0.18 ms with 32-bit ones, the same three steps as a dozen torch calls 0.17 ms, of host time mostly)"""
        rows = max(1, (grid.ny + strips - 1) // strips)
        n_strips = (grid.ny + rows - 1) // rows
        device = band.indices.device
        out = torch.empty(max(band.count, 1), dtype=torch.int32, device=device)
        scratch = torch.empty(3 * n_strips * grid.nz, dtype=torch.int32, device=device)
        _lib.check(_lib.lib.lsf_band_list_strip_major(band.pointer, band.count, ctypes.byref(grid), strips,
                                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                                                      dev.stream_ptr()), "lsf_band_list_strip_major")
        return dev.BandList(out, band.count, band.subset)

    def enqueue(self, i):
        f, lib, check = self.f, _lib.lib, _lib.check
        s_in, s_out = self.p_state[i % 2], self.p_state[(i + 1) % 2]
        gate = None if i < self.min_iterations else f.gate_ref(i - 1)
        rec = f.record_ptrs[i]
        none = ctypes.c_void_p(0)
        if self.fused_x:
            a, b = self.p_g
            raw = None
            for band in self.bands_first:
                check(lib.lsf_sobolev_state_gradient_x(s_in, self.p_canon, a, f.grid_ref, self.params_ref, self.p_taps,
                                                       self.n_taps, gate, rec, band.pointer, band.count,
                                                       int(self.boxes is not None), self.stream),
                      "lsf_sobolev_state_gradient_x")
            if self.boxes is not None:
                keep = self.last_iteration is None or i == self.last_iteration
                check(lib.lsf_sobolev_state_update_boxes(a, s_in, s_out, b if keep else none, f.grid_ref, self.params_ref,
                                                         self.p_taps, self.n_taps, gate, rec, self.p_boxes,
                                                         self.boxes[1], self.stream), "lsf_sobolev_state_update_boxes")
                return
            src, dst, axes = a, b, self.axes[1:-1]
        else:
            raw, a, b = self.p_g
            for band in self.bands:
                check(lib.lsf_sobolev_state_gradient(s_in, self.p_canon, raw, f.grid_ref, self.params_ref, gate, rec,
                                                     band.pointer, band.count, self.stream), "lsf_sobolev_state_gradient")
            src, dst, axes = raw, a, self.axes[:-1]
        # the FIRST pass takes the zero-preserving mask from the raw gradient (its own input) and leaves it as bits in its
        # output's fourth component; every later pass reads it there (no mask source: one load per voxel less)
        for axis in axes:
            for band in self.bands:
                check(lib.lsf_convolve_axis_listed4(src, dst, raw if src is raw else none, f.grid_ref, axis, self.p_taps,
                                                    self.n_taps, gate, band.pointer, band.count, self.stream),
                      "lsf_convolve_axis_listed4")
            src, dst = dst, (b if dst is a else a)
        if self.last_iteration is not None and i != self.last_iteration:
            dst = none
        for k, band in enumerate(self.bands_last):
            check(lib.lsf_sobolev_state_update(src, raw if src is raw else none, s_in, s_out, dst, f.grid_ref,
                                               self.params_ref, self.axes[-1],
                                               self.p_taps, self.n_taps, gate, rec, band.pointer, band.count,
                                               int(k == 0), self.stream), "lsf_sobolev_state_update")

    def final_gradient_planar(self, dims):
        """[c][z,]y,x float32 from the float4 buffer of the last executed iteration (API edge only)"""
        g4 = self.g4[self.final]
        return g4[..., :dims].movedim(-1, 0).contiguous()


class _SparseStateExceeded(Exception):
    """a call whose ping-pong states were initialised near the band only (dev.StatePrepare(sparse_reach=...)) met a warp
    update that can read beyond that; nothing of the caller's has been modified (SlavchevaEngine.optimize repeats the
    call on fully initialised states and keeps doing so for this optimizer)"""


# The ping-pong states are initialised only where an iteration can read them while every update stays below this many
# voxels (0 = everywhere, as before round 4), for volumes of at least SPARSE_MIN_VOXELS (below, a call is launch-bound and
# the initialisation passes cost nothing next to it)
# The library-enqueued call walks the INTERIOR band voxels box by box (lsf_slavcheva_state_iteration_boxes: neighbourhoods
# staged through LDS) instead of entry by entry when the listed voxels of the two ping-pong states -- 32 bytes each -- and
# what else an iteration touches crowd the 256 MB Infinity Cache: the list walk's 18 loads per voxel then miss to HBM and
# its L1s stand at their in-flight limit (profiles/r05_pmc_l2_tcp.txt), the box walk's four coalesced loads per 64 voxels
# do not: 120-125 against 138-144 us per 512^3 launch.  Measured on one box, list / box walk in us per launch
# (tools/box_kernel_ab.py, profiles/r05_box_walk_by_size.txt): sphere pairs 320^3 (84 MB of listed states) 43.6 / 43.6,
# 384^3 (122 MB) 65.2 / 62.2, 448^3 (169 MB) 103.1 / 88.8, 512^3 (224 MB) 138-144 / 119-125; the 512^3 depth pair (119 MB)
# 76.0 / 66.9.  Below ~100 MB the two are level (256^3: 30.8 us both) and the list walk needs no boxes built.
BOX_WALK_MIN_BAND_BYTES = 100 * 1000 * 1000
BOX_WALK_MIN_VOXELS = 1 << 25  # (volumes below this never reach the band size above: the boxes are not even counted)
SPARSE_REACH = int(os.environ.get("LSF_SPARSE_REACH", "2"))
SPARSE_MIN_VOXELS = int(os.environ.get("LSF_SPARSE_MIN_VOXELS", str(1 << 21)))


class SlavchevaEngine:
    """per-iteration-update optimizer with in-place re-warping of the live field
    (nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:332-408), D = 2 or 3, optionally on a z-slab."""

    def __init__(self, direct, level_set_term_enabled, sobolev_smoothing_enabled, data_term_method,
                 smoothing_term_method, gradient_descent_rate, data_term_weight, smoothing_term_weight,
                 isomorphic_enforcement_factor, level_set_term_weight, lower_threshold, upper_threshold,
                 max_iterations, min_iterations, sobolev_kernel, compute_energies=True, check_interval=32,
                 comm=None, use_band_list=True):
        self.direct = bool(direct)
        self.sobolev = bool(sobolev_smoothing_enabled)
        self.sobolev_kernel = sobolev_kernel
        if self.sobolev and sobolev_kernel is None:
            raise ValueError("sobolev_smoothing_enabled requires a sobolev_kernel")
        lam = float(isomorphic_enforcement_factor)
        # VECTORIZED ignores the Killing / level-set / thresholded options (slavcheva_optimizer2d.py:163-190)
        smoothing = smoothing_term_method if self.direct else _lib.SMOOTHING_TIKHONOV
        data = data_term_method if self.direct else _lib.DATA_BASIC
        level_set = bool(level_set_term_enabled) and self.direct
        energy = _lib.ENERGY_NONE if not compute_energies else \
            (_lib.ENERGY_DIRECT if self.direct else _lib.ENERGY_VECTORIZED)
        self.params = _lib.SlavchevaParams(lam, float(gradient_descent_rate), float(data_term_weight),
                                           float(smoothing_term_weight), float(level_set_term_weight), lam,
                                           float(np.float32(-2.0 * (1.0 + lam))), int(smoothing), int(data),
                                           int(level_set), int(energy), int(self.direct), 0)
        self.weights = (float(data_term_weight), float(smoothing_term_weight), float(level_set_term_weight))
        self.lo, self.hi = float(lower_threshold), float(upper_threshold)
        self.max_iterations, self.min_iterations = int(max_iterations), int(min_iterations)
        self.check_interval = max(1, int(check_interval))
        self.comm = comm
        self.use_band_list = bool(use_band_list)  # False: the fused kernel walks every voxel (measurements, tests)
        # whole-volume fixed-count calls are enqueued by the library in one piece (_optimize_run); False: the general path,
        # one foreign call per launch (tests hold the two against each other)
        self.library_run = True
        self.box_walk = None  # None: by band size (BOX_WALK_MIN_BAND_BYTES); True / False: always / never (tests)
        self.sobolev_boxes = True  # SobolevFusion on whole 3-D volumes: y pass, z pass and update box by box (False: lists)
        self.iteration_count = 0
        self.log = None
        self._gradient_state = None
        # opt-in per-iteration call-back f(level = 0, iteration, warp, gradient, max_warp) (device tensors, API layout),
        # where the reference writes its per-iteration visualisations (slavcheva_optimizer2d.py:387-388).  None: no cost.
        self.iteration_hook = None

    def _grid(self, live):
        if self.comm is not None and self.comm.active:
            L = self.comm.layout
            if L.axis == 1:
                # slabs cut along y: the launches name their voxels by band lists (every z), the grid carries the global
                # row of local row 0 (gather positions, reported indices) and the owned rows (energies)
                if live.dim() != 3 or live.shape[1] != L.n_local:
                    raise ValueError("y-slab runs need a 3-D local field with %d rows, got %r"
                                     % (L.n_local, tuple(live.shape)))
                g = dev.make_grid(live.shape)
                g.y_global_offset, g.ny_global = L.global_offset, L.n_global
                g.energy_y_begin, g.energy_y_end = L.begin, L.end
                return g
            if live.dim() != 3 or live.shape[0] != L.nz_local:
                raise ValueError("slab runs need a 3-D local field with %d slices, got %r"
                                 % (L.nz_local, tuple(live.shape)))
            return dev.make_grid(live.shape, L.z_begin, L.z_end, L.z_global_offset)
        return dev.make_grid(live.shape)

    def _slab(self):
        return self.comm is not None and self.comm.active

    def _gate_for(self, records, i):
        # iteration i runs iff i < min_iterations or (i < max_iterations and lo < max_warp[i-1] < hi)
        # (slavcheva_optimizer2d.py:360-362)
        return None if i < self.min_iterations else dev.make_gate(records, i - 1, _lib.GATE_SLAVCHEVA, self.lo,
                                                                  self.hi)

    def _enqueue_iteration(self, i, live_in, live_out, warp_in, warp_out, canonical, grid, records, gbufs, limit):
        """Sobolev path (planar fields): gradient kernel, zero-preserving separable filter, update + re-warp"""
        gate = self._gate_for(records, i)
        slab = self._slab()
        g0, t1, t2 = gbufs
        band = self._sobolev_band  # None: every voxel
        # z-slab: the list of the WHOLE local array serves the x / y passes (they also run on the halo slices), its owned
        # part everything else
        band_own = self._sobolev_band_owned if slab and band is not None else band
        dev.slavcheva_gradient(live_in, canonical, warp_in, g0, grid, self.params, gate, records, i, band_own)
        in_plane_grid = grid
        if slab:
            # the z pass of the filter reads len(kernel)//2 slices of the (x,y)-filtered field on either
            # side: exchange the raw gradient's halo once and run the x and y passes on the halo slices too
            self.comm.exchange_halos([g0])
            in_plane_grid = dev.make_grid(live_in.shape, 0, grid.nz, grid.z_global_offset)
        src, dst = g0, t1
        axes = _conv_axis_order(grid.dims)
        # on a band list the LAST pass runs in the launch of the update and the re-warp (the filtered gradient of a voxel is
        # all its update needs): four launches per iteration instead of five
        fuse_last = band is not None
        for axis in axes[:-1] if fuse_last else axes:
            dev.convolve_axis(src, dst, g0, grid if axis == 2 else in_plane_grid, axis, self.sobolev_kernel,
                              gate, band_own if axis == 2 else band)
            src, dst = dst, (t2 if dst is t1 else t1)
        if fuse_last:
            axis = axes[-1]
            dev.slavcheva_filter_update_rewarp(src, g0, live_in, dst, warp_out, live_out,
                                               grid if axis == 2 else in_plane_grid, self.params, axis,
                                               self.sobolev_kernel, gate, records, i, band_own if axis == 2 else band)
            src = dst
        else:
            dev.slavcheva_update_rewarp(live_in, canonical, src, warp_out, live_out, grid, self.params, gate,
                                        records, i, band_own)
        self._last_g = src
        if slab:
            self.comm.exchange_live_and_warp(live_out, warp_out)
            if i + 1 < limit and i + 1 >= self.min_iterations:
                self.comm.reduce_max(records, i)  # the next iteration's gate tests this record: make it global now

    def _enqueue_state_iteration(self, i, states, limit):
        """fused path: ONE kernel per iteration (and per band list) on the float4 state (live, u, v, w)"""
        f = self._fast
        s_in, s_out = f.p_state[i % 2], f.p_state[(i + 1) % 2]
        gate_ref = None if i < self.min_iterations else f.gate_ref(i - 1)
        run = _lib.lib.lsf_slavcheva_state_iteration
        if not self._slab():
            for band in f.bands:  # interior + boundary band voxels (or one list / the dense walk)
                status = run(s_in, f.p_canon, s_out, f.grid_ref, f.params_ref, gate_ref, f.record_ptrs[i],
                             band.pointer, band.count, band.subset, f.stream)
                if status:
                    _lib.check(status, "lsf_slavcheva_state_iteration")
            return
        # z-slab: what this iteration launches and whether the faces travel afterwards is planned in _plan_slab
        k = f.exchange_interval
        j = i % k
        exchange = j == k - 1 and i + 1 < limit
        resume = k > 1 and j == 0 and i > 0   # the iteration before this one left its exchange in flight
        if exchange:
            mode, (boundary, interior) = (_lib.SLAB_EXCHANGE_DEFERRED if k > 1 else _lib.SLAB_EXCHANGE), f.exchange_parts.get()
        elif resume:
            mode, (boundary, interior) = _lib.SLAB_RESUME, f.resume_parts.get()
        else:
            mode, (boundary, interior) = _lib.SLAB_LAUNCH, f.widened_parts[0 if j == k - 1 else k - 1 - j].get()
        if f.native is not None and getattr(f, "face_plan_args", None) is not None and \
                (exchange or j >= k - 2):
            # the face lists and the neighbours' face counts: made ONE iteration before the first exchange -- the host
            # enqueues an iteration in ~20 us, the card takes ~30, so that is where the host's lead over the card is
            # largest and the ~0.15 ms of host calls (a collective) starve it least (kernel trace of the loop-back,
            # round 4: planned behind the first iteration, with torch.sort for the merges, the card idled 0.4 ms there)
            (args, kwargs), f.face_plan_args = f.face_plan_args, None
            self._plan_compact_faces(f, *args, **kwargs)
        if f.native is not None and exchange and f.pending_face_plan is not None:
            self._finish_compact_faces(f)  # may fall back to the torch transport (slabs cut along y, neighbours disagree)
        if f.native is not None:  # the whole iteration in one host call (lsf_slab.hip): RCCL on the library's stream
            status = _lib.lib.lsf_slab_state_iteration(f.native, s_in, f.p_canon, s_out, f.layout_ref, boundary.array,
                                                       boundary.n, interior.array, interior.n, f.params_ref, gate_ref,
                                                       f.record_ptrs[i], mode, f.faces_ref, f.stream)
            if status:
                _lib.check(status, "lsf_slab_state_iteration")
        else:
            # torch.distributed transport (gloo tests, fallback), the same schedule: boundary slices first, then the halo
            # exchange on a second stream WHILE the interior runs -- and, in an exchange group, while the next
            # iteration's halo-independent part runs
            main = torch.cuda.current_stream()
            for grid_ref, bands in boundary.launches:
                for band in bands:
                    _lib.check(run(s_in, f.p_canon, s_out, grid_ref, f.params_ref, gate_ref, f.record_ptrs[i],
                                   band.pointer, band.count, band.subset, f.stream), "lsf_slavcheva_state_iteration")
            if exchange:
                boundary_done, halos_done = self._events[i % 2]
                boundary_done.record(main)
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(boundary_done)
                    self.comm.exchange_state(states[(i + 1) % 2])
                    halos_done.record(self._comm_stream)
                self._pending_halos = halos_done
            if resume and self._pending_halos is not None:
                main.wait_event(self._pending_halos)
                self._pending_halos = None
            for grid_ref, bands in interior.launches:
                for band in bands:
                    _lib.check(run(s_in, f.p_canon, s_out, grid_ref, f.params_ref, gate_ref, f.record_ptrs[i],
                                   band.pointer, band.count, band.subset, f.stream), "lsf_slavcheva_state_iteration")
            if exchange and k == 1:
                main.wait_event(self._pending_halos)
                self._pending_halos = None
        if i + 1 < limit and i + 1 >= self.min_iterations:
            self.comm.reduce_max(f.records, i)  # the next iteration's gate tests this record: make it global now

    def _plan_compact_faces(self, f, live, bands, cut, lo, hi, lo_rank, hi_rank, faces=None):
        """Only the band voxels of a face travel: every other voxel of the boundary slices never changes (and both ranks
        hold it already).  The sender gathers state[its boundary band voxels], the receiver scatters into its halo band
        voxels -- the same physical voxels in the same ascending order, because both ranks cut their lists out of
        identical initial data; the counts are cross-checked with the neighbours once, and whole slices travel if any
        rank disagrees (a caller that hands inconsistent halos)."""
        L = self.comm.layout
        h = L.halo
        # Host work between two launches (the card waits for it: kernel trace of the loop-back): a face
        # is (device pointer, count) -- a run of a sorted list is pointer arithmetic, no tensor views -- and the faces
        # that have entries in both lists are merged in ONE launch into ONE buffer
        if getattr(self, "_no_face", None) is None or self._no_face.device != live.device:
            self._no_face = torch.zeros(4, dtype=torch.int32, device=live.device)  # a valid address for an empty face
        none = (self._no_face.data_ptr(), 0)
        keep = [self._no_face]
        if faces is not None:  # slabs cut along y: the caller filtered the four lists out by row
            def pad(e):
                if e is None or e[1] == 0:
                    return none
                keep.append(e[0])
                return e[0].data_ptr(), int(e[1])
            send, recv = [pad(e) for e in faces["send"]], [pad(e) for e in faces["recv"]]
        else:
            base = [b.indices.data_ptr() for b in bands]
            merges = []  # (run a, run b, offset into the merged buffer): ascending merge of the INTERIOR and the BOUNDARY
                         # entries of a face (lsf_merge_sorted_runs)
            merged_words = 0

            def union(z0, z1):
                nonlocal merged_words
                runs = [(p + 4 * c[z0], c[z1] - c[z0]) for p, c in zip(base, cut) if c[z1] > c[z0]]
                if not runs:
                    return none
                if len(runs) == 1:
                    return runs[0]
                merges.append((runs[0], runs[1], merged_words))
                merged_words += runs[0][1] + runs[1][1]
                return None, runs[0][1] + runs[1][1], len(merges) - 1  # its address follows below
            plan = [union(L.z_begin, L.z_begin + h) if lo else none, union(L.z_end - h, L.z_end) if hi else none,
                    union(L.z_begin - h, L.z_begin) if lo else none, union(L.z_end, L.z_end + h) if hi else none]
            keep += [b.indices for b in bands]
            if merges:
                n = len(merges)
                merged = torch.empty(merged_words, dtype=torch.int32, device=live.device)
                keep.append(merged)
                out = [merged.data_ptr() + 4 * m[2] for m in merges]
                plan = [e if e[0] is not None else (out[e[2]], e[1]) for e in plan]
                vp, i64 = ctypes.c_void_p * n, ctypes.c_int64 * n
                _lib.check(_lib.lib.lsf_merge_sorted_runs(vp(*[m[0][0] for m in merges]), i64(*[m[0][1] for m in merges]),
                                                          vp(*[m[1][0] for m in merges]), i64(*[m[1][1] for m in merges]),
                                                          vp(*out), n, dev.stream_ptr()), "lsf_merge_sorted_runs")
            send, recv = plan[:2], plan[2:]
        # A rank's halo holds the neighbour's boundary slices, so what it expects to receive IS what the neighbour sends --
        # if the caller cut consistent slabs.  That contract is cross-checked with the neighbours on EVERY call:
        # mismatched message sizes would hang or corrupt the transport, and whether to check cannot depend on anything
        # one rank alone sees (a rank whose data changed would enter the collective alone).  The counts are host numbers
        # (cut positions), so the collective is STARTED here and its result is READ when the first exchange is enqueued.
        # On the native transport it is the library's own (lsf_slab_face_counts_begin / _end: an ncclAllGather on the
        # communicator's stream, ~15 us of host time; through torch.distributed the pinned copies, the collective and
        # the event cost ~0.15 ms of host calls, which the card spent idle).  (Checking on an optimizer's first call only
        # measured 2.47 against 2.55 ms per slab call, profiles/r04_slab_rccl_loopback.txt: a hang is worse.)
        check = None
        counts = [send[0][1], send[1][1], recv[0][1], recv[1][1]]
        world = torch.distributed.get_world_size(self.comm.group)
        if f.native is not None and not self.comm.stage_through_host:
            _lib.check(_lib.lib.lsf_slab_face_counts_begin(f.native, (ctypes.c_int64 * 4)(*counts)),
                       "lsf_slab_face_counts_begin")
            check = ("native", None)
        elif self.comm.stage_through_host:  # gloo (tests): a host collective, done at once
            mine = torch.tensor(counts, dtype=torch.int64)
            rows = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(rows, mine, group=self.comm.group)
            check = (torch.stack(rows), None)
        else:
            if getattr(self, "_plan_stream", None) is None or self._plan_stream.device != live.device:
                self._plan_stream = torch.cuda.Stream(device=live.device)
            with torch.cuda.stream(self._plan_stream):
                staged = dev.pinned_scratch("face counts out", 4, torch.int64)
                staged.copy_(torch.tensor(counts, dtype=torch.int64))
                mine = staged.to(live.device, non_blocking=True)
                rows = [torch.empty_like(mine) for _ in range(world)]
                torch.distributed.all_gather(rows, mine, group=self.comm.group)
                landed = dev.pinned_scratch("face counts in", 4 * world, torch.int64)
                landed.copy_(torch.cat(rows), non_blocking=True)
                done = torch.cuda.Event()
                done.record()
            check = (landed.view(world, 4), done)
        f.pending_face_plan = (send, recv, check, live.device, keep)

    def _finish_compact_faces(self, f):
        """second half of _plan_compact_faces, when the first exchange is enqueued: the neighbours' counts (the collective
        started with the launch plan has long finished), then the lsf_slab_faces descriptor -- or, if a neighbour
        disagrees, whole faces (z-slabs) / the torch transport with its packed staging buffers (slabs cut along y)"""
        send, recv, check, device, keep = f.pending_face_plan
        f.pending_face_plan = None
        ok = True
        if check is not None:
            table, done = check
            if done is not None:
                done.synchronize()
            if isinstance(table, str):  # the library's collective
                world = self.comm.native_identity()[1]
                flat = (ctypes.c_int64 * (4 * world))()
                _lib.check(_lib.lib.lsf_slab_face_counts_end(f.native, flat), "lsf_slab_face_counts_end")
                rows = [list(flat[4 * r:4 * r + 4]) for r in range(world)]
            else:
                rows = table.tolist()
            # every rank sees every row, so all ranks reach the same verdict without a second collective: a rank's lower
            # boundary lands in its lower neighbour's UPPER halo, its upper boundary in the upper neighbour's LOWER halo
            if len(rows) == 1:  # the one-GPU loop-back: this rank is its own neighbour on both sides
                ok = rows[0][0] == rows[0][3] and rows[0][1] == rows[0][2]
            else:
                for r in range(len(rows) - 1):
                    ok &= rows[r][1] == rows[r + 1][2] and rows[r + 1][0] == rows[r][3]
            self._faces_verified = ok
        if not ok:
            import warnings
            if self.comm.layout.axis == 1:
                warnings.warn("slab halos are not consistent with the neighbours' slabs (band voxel counts differ): the "
                              "rows travel whole through torch.distributed")
                f.native = None
                if not hasattr(self, "_comm_stream"):
                    self._comm_stream = torch.cuda.Stream(device=device)
                    self._events = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]
            else:
                warnings.warn("slab halos are not consistent with the neighbours' slabs (band voxel counts differ): whole "
                              "slices are exchanged")
            return
        faces = _lib.SlabFaces()
        entries = [(name, side, e) for side in range(2) for name, e in (("send", send[side]), ("recv", recv[side]))]
        msgs = torch.empty(4 * sum(max(e[1], 1) for _, _, e in entries), dtype=torch.float32, device=device)  # all four
        f.face_tensors = keep + [msgs]
        at = msgs.data_ptr()
        for name, side, (pointer, count) in entries:
            getattr(faces, name + "_list")[side] = pointer
            getattr(faces, name + "_msg")[side] = at
            getattr(faces, name + "_count")[side] = count
            at += 16 * max(count, 1)
        f.faces = faces
        f.faces_ref = ctypes.byref(faces)

    class _Parts:
        """the launches of one phase of a slab iteration: (grid, band lists) pairs, also as a ctypes lsf_slab_part array"""

        def __init__(self, launches):
            self.launches = [(ctypes.byref(g), bands) for g, bands in launches]
            self._grids = [g for g, _ in launches]
            self.n = len(launches)
            self.array = (_lib.SlabPart * max(self.n, 1))()
            for k, (g, bands) in enumerate(launches):
                self.array[k].grid = g
                self.array[k].n_lists = len(bands)
                for j, band in enumerate(bands):
                    self.array[k].band_list[j] = band.pointer.value or None
                    self.array[k].band_count[j] = band.count
                    self.array[k].band_subset[j] = band.subset

    def _slab_cut_slices(self, grid):
        """the slices at which a z-slab run cuts its band lists (ascending): every boundary of a widened, boundary,
        interior or resume range"""
        L = self.comm.layout
        h = L.halo
        zs = sorted({z for e in range(h + 1) for z in (L.z_begin - e, L.z_end + e)} |
                    {L.z_begin + h, L.z_end - h, L.z_begin + 1, L.z_end - 1})
        return [z for z in zs if 0 <= z <= grid.nz]

    def _plan_slab(self, f, live, grid, bands, limit, prepared=None):
        """Launch plan of a z-slab rank (fused path).  ONE band list of the whole local array (owned slices + halos) is
        cut by z -- it is sorted, so every z-range is a contiguous run of it.
        Exchange groups: with a halo of h slices and a fixed iteration count the faces travel only every h-th iteration
        (an RCCL send / recv costs ~50 us of latency whatever its size, a 256^3 iteration 40 us): iteration j of a group
        runs over the owned range WIDENED by h - 1 - j slices on every interior side -- it recomputes what the
        neighbour computes for those slices, bit for bit, from inputs that are still valid: every iteration consumes
        one slice of validity (stencils reach 1 slice, the re-warp gather floor(|w_z|) + 1 = 1 while updates stay
        below one voxel; the guard in optimize() enforces that) -- and only the last iteration of a group splits into
        boundary slices -> exchange || interior.  Energies count owned slices only (lsf_grid::energy_z_*).  Gated runs
        (the stop test can fire) exchange and reduce every iteration."""
        L = self.comm.layout
        if L.axis == 1:
            return self._plan_slab_y(f, live, grid, bands, limit)
        h = L.halo
        lo, hi = L.rank > 0, L.rank < L.world - 1
        _, _, lo_rank, hi_rank = self.comm.native_identity()
        lo, hi = lo or lo_rank >= 0, hi or hi_rank >= 0
        own = L.z_end - L.z_begin
        if own < 2 * h:
            raise ValueError("a slab of %d slices is too thin for a %d-slice halo" % (own, h))
        fixed = self.min_iterations >= limit
        f.exchange_interval = h if fixed and not getattr(self, "_exchange_every_iteration", False) else 1
        slice_voxels = grid.ny * grid.nx
        listed = bands[0].indices is not None
        if listed and prepared is not None:  # the prepare pass brought the positions of the z cuts along
            zs = self._slab_cut_slices(grid)
            cut = [dict(zip(zs, [prepared.cut_totals[b.subset] if z == grid.nz else c
                                 for z, c in zip(zs, prepared.cuts[b.subset])])) for b in bands]
        elif listed:  # positions of the z cuts inside every list: one searchsorted per list, one host read
            zs = self._slab_cut_slices(grid)
            keys = torch.tensor([z * slice_voxels for z in zs], dtype=torch.int32, device=live.device)
            cuts = torch.stack([torch.searchsorted(b.indices[:b.count], keys) if b.count else torch.zeros_like(keys,
                               dtype=torch.int64) for b in bands]).cpu().tolist()
            cut = [dict(zip(zs, c)) for c in cuts]

        def grid_of(z0, z1):
            g = dev.make_grid(live.shape, z0, z1, grid.z_global_offset)
            g.energy_z_begin, g.energy_z_end = L.z_begin, L.z_end
            return g

        def lists_of(ranges):
            """the band lists covering the z-ranges (ascending, disjoint): views of the global lists, concatenated when
            there is more than one range; at least one (possibly empty) list so that the launch still reports"""
            if not listed:
                return None
            out = []
            for b, c in zip(bands, cut):
                pieces = [b.indices[c[z0]:c[z1]] for z0, z1 in ranges if c[z1] > c[z0]]
                if pieces:
                    idx = pieces[0] if len(pieces) == 1 else torch.cat(pieces)
                    out.append(dev.BandList(idx, idx.numel(), b.subset))
            return out or [dev.BandList(bands[0].indices[:1], 0, bands[0].subset)]

        def parts(ranges):
            ranges = [r for r in ranges if r[1] > r[0]]
            if not ranges:
                return SlavchevaEngine._Parts([])
            if listed:  # one launch per subset over all ranges
                return SlavchevaEngine._Parts([(grid_of(ranges[0][0], ranges[-1][1]), lists_of(ranges))])
            return SlavchevaEngine._Parts([(grid_of(z0, z1), [dev.BandList.none()]) for z0, z1 in ranges])

        # Every (boundary part, interior part) pair is built when an iteration first asks for it: only the first
        # iteration's pair stands between the list sizes and the first launch, the others are made while launches are
        # already queued (13 descriptors, ~0.1 ms of host work at 256^3)
        empty = SlavchevaEngine._Parts([])
        f.widened_parts = [_Lazy(lambda e=e: (empty, parts([(L.z_begin - (e if lo else 0), L.z_end + (e if hi else 0))])))
                           for e in range(f.exchange_interval)]
        z_lo, z_hi = L.z_begin + (h if lo else 0), L.z_end - (h if hi else 0)
        # (measured and left alone, profiles/r04_slab_rccl_loopback.txt: no boundary-first split -- the whole owned range in
        # one launch, the exchange hidden behind the next iteration's halo-independent part only -- is within 2 %)
        f.exchange_parts = _Lazy(lambda: (parts(([(L.z_begin, z_lo)] if lo else []) +
                                                ([(z_hi, L.z_end)] if hi else [])), parts([(z_lo, z_hi)])))
        # first iteration of a group, while the previous group's exchange may still be in flight: the owned slices that
        # do not touch a halo slice first, the rest (the widened range's outer slices) after the halos have arrived
        e_last = f.exchange_interval - 1
        in_lo, in_hi = L.z_begin + (1 if lo else 0), L.z_end - (1 if hi else 0)
        f.resume_parts = _Lazy(lambda: (parts([(in_lo, in_hi)]),
                                        parts(([(L.z_begin - e_last, in_lo)] if lo else []) +
                                              ([(in_hi, L.z_end + e_last)] if hi else []))))
        self._pending_halos = None
        f.native = self.comm.native()
        f.faces_ref = None
        if f.native is not None:
            f.layout = _lib.SlabLayoutC(grid.nz, grid.ny, grid.nx, L.z_begin, L.z_end, h, lo_rank, hi_rank)
            f.layout_ref = ctypes.byref(f.layout)
            # compact faces are planned when the first exchange is enqueued (_enqueue_state_iteration): the plan costs a
            # collective and a host read (~0.2 ms) that then wait behind the iterations already queued, not in front of them
            f.pending_face_plan = f.face_plan_args = None
            if listed and os.environ.get("LSF_SLAB_FACES", "compact") != "full":
                f.face_plan_args = ((live, bands, cut, lo, hi, lo_rank, hi_rank), {})
        elif not hasattr(self, "_comm_stream"):
            self._comm_stream = torch.cuda.Stream(device=live.device)
            self._events = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]

    def _plan_slab_y(self, f, live, grid, bands, limit):
        """_plan_slab for slabs cut along Y (SlabLayout(axis=1)): the same schedule -- exchange groups of h iterations over
        row ranges widened by h - 1 - j rows, boundary rows -> exchange || interior rows, deferred waits -- with one
        difference: a row range is not a contiguous run of the sorted band list, so every part is its own list, filtered
        out of the lists of the whole local array once per call (y = (index / nx) mod ny: two integer operations and a
        compaction per part, on the device).  Faces are rows: nz runs of h * nx float4, which only ever travel compacted
        to their band voxels (gather / scatter by list: a strided face costs nothing extra), or through packed staging
        buffers on the torch transport."""
        L = self.comm.layout
        h = L.halo
        lo, hi = L.rank > 0, L.rank < L.world - 1
        _, _, lo_rank, hi_rank = self.comm.native_identity()
        lo, hi = lo or lo_rank >= 0, hi or hi_rank >= 0
        own = L.end - L.begin
        if own < 2 * h:
            raise ValueError("a slab of %d rows is too thin for a %d-row halo" % (own, h))
        if bands[0].indices is None:
            raise ValueError("slabs cut along y run on band lists (use_band_list=True)")
        fixed = self.min_iterations >= limit
        f.exchange_interval = h if fixed and not getattr(self, "_exchange_every_iteration", False) else 1
        nx, ny = grid.nx, grid.ny
        rows = [((b.indices[:b.count] // nx) % ny) if b.count else None for b in bands]

        def lists_of(ranges):
            out = []
            for b, y in zip(bands, rows):
                if y is None:
                    continue
                keep = None
                for y0, y1 in ranges:
                    m = (y >= y0) & (y < y1)
                    keep = m if keep is None else keep | m
                idx = b.indices[:b.count][keep].contiguous()
                if idx.numel():
                    out.append(dev.BandList(idx, idx.numel(), b.subset))
            return out or [dev.BandList(bands[0].indices[:1], 0, bands[0].subset)]

        def parts(ranges):
            ranges = [r for r in ranges if r[1] > r[0]]
            if not ranges:
                return SlavchevaEngine._Parts([])
            return SlavchevaEngine._Parts([(grid, lists_of(ranges))])  # the grid already carries the owned rows (energies)

        empty = SlavchevaEngine._Parts([])
        f.widened_parts = [_Lazy(lambda e=e: (empty, parts([(L.begin - (e if lo else 0), L.end + (e if hi else 0))])))
                           for e in range(f.exchange_interval)]
        y_lo, y_hi = L.begin + (h if lo else 0), L.end - (h if hi else 0)
        f.exchange_parts = _Lazy(lambda: (parts(([(L.begin, y_lo)] if lo else []) + ([(y_hi, L.end)] if hi else [])),
                                          parts([(y_lo, y_hi)])))
        e_last = f.exchange_interval - 1
        in_lo, in_hi = L.begin + (1 if lo else 0), L.end - (1 if hi else 0)
        f.resume_parts = _Lazy(lambda: (parts([(in_lo, in_hi)]),
                                        parts(([(L.begin - e_last, in_lo)] if lo else []) +
                                              ([(in_hi, L.end + e_last)] if hi else []))))
        self._pending_halos = None
        f.native = self.comm.native()
        f.faces_ref = None
        f.pending_face_plan = f.face_plan_args = None
        if f.native is not None:
            # the library's exchange only ever sees compacted faces here; its layout argument is validated, not used
            f.layout = _lib.SlabLayoutC(grid.ny, grid.nz, grid.nx, L.begin, L.end, h, lo_rank, hi_rank)
            f.layout_ref = ctypes.byref(f.layout)

            def face(y0, y1):
                got = lists_of([(y0, y1)])
                if len(got) == 1:
                    return got[0].indices[:got[0].count] if got[0].count else got[0].indices[:0], got[0].count
                idx = torch.sort(torch.cat([g.indices[:g.count] for g in got])).values.contiguous()
                return idx, idx.numel()
            f.face_plan_args = ((live, None, None, lo, hi, lo_rank, hi_rank),
                                dict(faces=dict(send=[face(L.begin, L.begin + h) if lo else None,
                                                      face(L.end - h, L.end) if hi else None],
                                                recv=[face(L.begin - h, L.begin) if lo else None,
                                                      face(L.end, L.end + h) if hi else None])))
        if f.native is None and not hasattr(self, "_comm_stream"):
            self._comm_stream = torch.cuda.Stream(device=live.device)
            self._events = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]

    def optimize(self, live, canonical, finalize=None):
        """live, canonical: float32 device tensors (z-slab runs: the local slab incl. halos).  Returns a
        SlavchevaOutcome holding the final fields on the device; the caller's tensors are not modified.

        z-slab runs never abort on large warps (the reference only stops at 10 000 voxels,
        slavcheva_optimizer2d.py:360-362): an iteration is exact on a slab while its re-warp gather, floor(|w_z|) + 1
        slices, stays inside what the halo schedule keeps valid -- one slice inside an exchange group, the halo width
        with an exchange per iteration.  Every rank sees the same (reduced) maxima, so when a batch breaks that bound
        all ranks together discard the call's work and run it again from its inputs on a WIDER internal slab: halo
        ceil(max) + 1 (live and canonical slices fetched from the neighbours), the faces exchanged every iteration
        ("fall back to a wider exchange if the max exceeds 1").  The result is bit for bit the
        whole-volume one (tests/test_gpu_slab_many_ranks.py)."""
        if not self._slab():
            try:
                return self._optimize(live, canonical, finalize)
            except _SparseStateExceeded:
                # an update of SPARSE_REACH voxels or more may have gathered from a part of the states that was never
                # initialised.  The finalize pass has left the caller's tensors alone (its skip flag): the call runs again
                # on fully initialised states, and so does every later call of this optimizer (its data move that far)
                self._sparse_disabled = True
                torch.cuda.synchronize()
                return self._optimize(live, canonical, finalize)
        self._slab_restore = None
        try:
            outcome = self._optimize(live, canonical, finalize)
            self._slab_restore = None
            return outcome
        except _HaloTooNarrow as exc:
            torch.cuda.synchronize()  # nothing of the abandoned attempt (an exchange left in flight) may linger
            if self._slab_restore is not None:  # the abandoned attempt's finalize pass has written the caller's tensor
                self._slab_restore[0].copy_(self._slab_restore[1])
                self._slab_restore = None
            return self._optimize_widened(live, canonical, exc.max_update)

    def _optimize_widened(self, live, canonical, max_update):
        import copy
        import math
        from .slab import SlabComm, SlabLayout
        L = self.comm.layout
        ax = L.axis  # 0: z-slabs, 1: slabs cut along y -- the same procedure along that axis
        per = L.z1 - L.z0
        while True:
            h2 = max(L.halo, int(math.floor(max_update)) + 2)
            if 2 * h2 > per:  # the boundary / interior split of a slab iteration needs two disjoint boundary ranges
                raise RuntimeError("warp update of %.3f voxels needs a %d-slice halo, more than half a slab of %d "
                                   "slices: use fewer, thicker slabs" % (max_update, h2, per))
            L2 = SlabLayout(L.nz_global, L.rank, L.world, h2, axis=ax)
            # the same kind of communicator on the wider layout; it BORROWS the library-side RCCL communicator (which knows
            # ranks, not layouts: every call names its layout), so the re-run keeps the one-host-call-per-iteration
            # transport instead of ~150 us of torch.distributed point-to-point per iteration
            comm2 = type(self.comm)(L2, self.comm.group)
            comm2._native = self.comm.native()
            wide = []
            for t in (live, canonical):
                shape = list(t.shape)
                shape[ax] = L2.n_local
                w = torch.empty(shape, dtype=t.dtype, device=t.device)
                w.narrow(ax, L2.begin, L2.end - L2.begin).copy_(t.narrow(ax, L.begin, L.end - L.begin))
                wide.append(w)
            comm2.exchange_halos(wide)
            clone = copy.copy(self)
            clone.comm = comm2
            clone._exchange_every_iteration = True
            for cached in ("_faces_verified", "_cut_chunk_cache", "_fast"):
                clone.__dict__.pop(cached, None)
            try:
                outcome = clone._optimize(wide[0], wide[1], None)
                break
            except _HaloTooNarrow as exc:  # a later iteration moved further still
                torch.cuda.synchronize()
                max_update = max(max_update + 1.0, exc.max_update)
        self.iteration_count, self.log = clone.iteration_count, clone.log
        off = L2.halo_lo - L.halo_lo
        window = slice(off, off + L.nz_local)
        self._gradient_state = ("wide", clone, window, ax)
        grid = self._grid(live)
        if outcome.state is not None:
            return SlavchevaOutcome(grid, canonical, state=outcome.state.narrow(ax, off, L.n_local).contiguous())
        return SlavchevaOutcome(grid, canonical, live=outcome.live().narrow(ax, off, L.n_local).contiguous(),
                                warp_planar=outcome.warp_planar().narrow(1 + ax, off, L.n_local).contiguous())

    def _optimize(self, live, canonical, finalize=None):
        """one attempt of optimize() (see there)
        finalize = (live_out, lower_threshold, statistics): the arguments the caller is going to pass to
        outcome.finalize() -- with a fixed iteration count and a whole volume (no stop test can fire) the finalize pass is then enqueued
        right behind the last iteration and the records and statistics are read with ONE host synchronisation."""
        if live.shape != canonical.shape:
            raise ValueError("live and canonical fields must have the same shape")
        # what the previous call left for gradient_field() and its launcher still holds that call's ping-pong states: let
        # go of them BEFORE this call allocates its own, so that the allocator hands the same blocks out again (otherwise
        # the footprint doubles and the first three calls of an optimizer each pay device allocations: 120 / 131 / 70 ms
        # against 9 ms at 512^3, tools/step_times.py)
        self._gradient_state = None
        self._fast = None
        grid = self._grid(live)
        dims = grid.dims
        n_rec = max(self.max_iterations, self.min_iterations, 1)
        slab = self._slab()
        if (self.library_run and finalize is not None and not slab and not self.sobolev and self.use_band_list
                and self.iteration_hook is None
                and self.min_iterations > 0 and self.min_iterations >= self.max_iterations
                and dev.buffer_addressing_ok(grid)):
            # a whole volume, a fixed iteration count, no Sobolev filter, nobody watching the iterations: the whole call is
            # enqueued by the library (two host calls; slavcheva_optimizer2d.py:354-388's loop without a Python iteration)
            return self._optimize_run(live, canonical, grid, finalize)
        if slab:
            need = 1 if not self.sobolev else max(1, len(self.sobolev_kernel) // 2)
            if self.comm.layout.halo < need:
                raise ValueError("slab halo of %d slices is too narrow: this configuration needs >= %d"
                                 % (self.comm.layout.halo, need))
        prepared = None
        sparse = False
        warp_zeroed = None
        # SobolevFusion on band lists of a whole volume runs on the float4 layouts too (one vector-memory instruction per
        # neighbour / tap instead of one per component: lsf_sobolev_state.hip); z-slabs, filters of other lengths and
        # list-less runs keep the planar kernels
        sob_state = (self.sobolev and self.use_band_list and not slab and dev.buffer_addressing_ok(grid)
                     and len(self.sobolev_kernel) in dev.LISTED_TAP_COUNTS)
        planar_sobolev = self.sobolev and not sob_state
        fused_prepare = not planar_sobolev and self.use_band_list and dev.buffer_addressing_ok(grid)
        if fused_prepare:
            # one pass: both states + the INTERIOR / BOUNDARY band lists of the WHOLE local array.  Launched first: the
            # host sets up records and launch arguments while it runs, and only then waits for the list sizes -- and, in
            # a z-slab run, for the positions of the z cuts in the lists (slices of a multiple of 1024 voxels: the
            # prepare pass's per-chunk prefix counts hold them)
            cut_chunks = None
            if slab and self.comm.layout.axis == 0 and (grid.ny * grid.nx) % dev.StatePrepare.CHUNK == 0:
                zs = self._slab_cut_slices(grid)
                key = (tuple(zs), grid.ny * grid.nx, live.device)
                if getattr(self, "_cut_chunk_cache", (None, None))[0] != key:
                    per_slice = grid.ny * grid.nx // dev.StatePrepare.CHUNK
                    self._cut_chunk_cache = (key, torch.tensor([z * per_slice for z in zs], dtype=torch.int64,
                                                               device=live.device))
                cut_chunks = self._cut_chunk_cache[1]
            # whole volumes: the states are initialised near the band only (a quarter of the voxels of a 256^3 sphere
            # pair), valid while every update stays below SPARSE_REACH voxels -- checked on the device in front of an
            # early finalize pass and on the host behind every batch
            # Slabs: in exchange groups only (a fixed iteration count; an update of one voxel or more already sends the
            # call to _optimize_widened, which exchanges every iteration and runs on full states), with a halo of at
            # least SPARSE_REACH slices: then every chunk a rank reads in its halo is one the owner initialises too (the
            # band voxel that makes it needed lies inside the owner's halo), so even whole faces carry valid data
            slab_groups = (slab and self.min_iterations >= max(self.max_iterations, self.min_iterations)
                           and not getattr(self, "_exchange_every_iteration", False)
                           and self.comm.layout.halo >= max(SPARSE_REACH, 2))
            sparse = ((not slab or slab_groups) and SPARSE_REACH > 0 and dev.n_voxels(grid) >= SPARSE_MIN_VOXELS
                      and self.iteration_hook is None and not getattr(self, "_sparse_disabled", False))
            # the listed finalize pass wants a zero-filled warp output (192 MB at 256^3, 26 us): filled in the call's
            # prologue, where the card waits for the host, instead of behind the last iteration.  Up to 256^3 IN FRONT of
            # the counting pass: the states written behind it are then the last thing to pass through the 256 MB Infinity
            # Cache before the first two iterations read them (filled behind the states it evicted them: 1.846-1.858
            # against 1.817-1.825 ms per step, three alternating runs on one box); a larger volume's fill (1.6 GB at
            # 512^3) would only keep the list sizes from the host (8.67 against 8.53 ms)
            fill_first = dev.n_voxels(grid) <= (1 << 24)
            if finalize is not None and not slab and fill_first:
                warp_zeroed = torch.zeros(tuple(live.shape) + (dims,), dtype=torch.float32, device=live.device)
            self._sparse_used = sparse  # (tests and measurements look at this)
            prepared = dev.StatePrepare(live, canonical, dev.full_range(grid), cut_chunks,
                                        sparse_reach=SPARSE_REACH if sparse else 0)
            if finalize is not None and not slab and not fill_first:
                warp_zeroed = torch.zeros(tuple(live.shape) + (dims,), dtype=torch.float32, device=live.device)
        live_at_entry = None
        if slab and finalize is not None and finalize[0] is not None and not planar_sobolev \
                and self.min_iterations >= max(self.max_iterations, self.min_iterations):
            # the finalize pass of a fixed-count slab call is enqueued behind the last iteration, before the records of
            # every rank have said whether the call stands: what it overwrites is kept (a copy while the card waits for
            # the host anyway) -- instead of a launch, a synchronisation and a read-back behind the records
            live_at_entry = finalize[0].clone()
        records = dev.new_records(n_rec, live.device)
        self._last_g = None
        lives = warps = gbufs = states = sob = None
        if planar_sobolev:
            lives = [live.clone(), live.clone()]
            warps = [torch.zeros((dims,) + tuple(live.shape), dtype=torch.float32, device=live.device)
                     for _ in range(2)]
            gbufs = [torch.zeros_like(warps[0]) for _ in range(3)]
            # Band list: the gradient is zero outside the narrow band and the zero-preserving filter keeps it there
            # (math_utils/convolution.py:118-127), so gradient, filter passes and update visit band voxels only; the
            # zero-initialised g buffers and the two (live, 0) sets hold everything else.  z-slab runs list the whole
            # local array (the x / y passes also run on the halo slices) and cut the owned part out of that list: it
            # is sorted, so the owned slices are one contiguous run of it.
            self._sobolev_band = self._sobolev_band_owned = None
            if self.use_band_list and len(self.sobolev_kernel) in dev.LISTED_TAP_COUNTS:
                self._sobolev_band = dev.band_list(live, canonical, dev.full_range(grid), _lib.BAND_ALL)
                if slab:
                    b = self._sobolev_band
                    slice_voxels = grid.ny * grid.nx
                    keys = torch.tensor([grid.z_begin * slice_voxels, grid.z_end * slice_voxels], dtype=torch.int32,
                                        device=live.device)
                    lo, hi = torch.searchsorted(b.indices[:b.count], keys).tolist() if b.count else (0, 0)
                    self._sobolev_band_owned = dev.BandList(b.indices[lo:hi] if hi > lo else b.indices[:1], hi - lo,
                                                            b.subset)
        else:
            # Both ping-pong states start as (live, 0): the fused kernel only visits the voxels of the band list and
            # the rest must already hold their final values (lsf_slavcheva_state_iteration); slab halos start valid.
            whole = dev.full_range(grid)
            listed = None
            states = prepared.states if fused_prepare else dev.state_pack(live, None, grid, copies=2)
            n = dev.n_voxels(grid)
            f = dev.IterationLauncher(grid, records, _lib.GATE_SLAVCHEVA, self.lo, self.hi)
            f.p_state = [f.pointer(t, 4 * n, "state") for t in states]
            f.p_canon = f.pointer(canonical, n, "canonical")
            f.params_ref = ctypes.byref(self.params)
            f.stream = dev.stream_ptr()  # the launch stream of this call (one ctypes object, not one per launch)
            f.native = None
            if fused_prepare:
                bands, unlisted = prepared.collect()
                # z-slab runs: the lists cover the whole local array (owned slices + halos), like the dense finalize
                # pass does; the unlisted counts would too, so statistics (never asked for there) take the dense pass
                listed = (live, bands, None if slab else unlisted)
            else:
                bands = dev.band_lists(live, canonical, whole) if self.use_band_list else [dev.BandList.none()]
            f.bands = bands
            self._fast = f
            if sob_state:
                g4 = [torch.zeros(tuple(live.shape) + (4,), dtype=torch.float32, device=live.device)
                      for _ in range(2 if _SobolevStatePlan.fuses_x(grid) else 3)]
                n_max = max(self.max_iterations, self.min_iterations)
                every = self.iteration_hook is not None or self.min_iterations < n_max
                # whole 3-D volumes of whole boxes: everything behind the x pass box by box in one launch
                boxes = None
                if (fused_prepare and not slab and self.sobolev_boxes and _SobolevStatePlan.fuses_x(grid)
                        and dev.boxes_ok(grid) and dev.n_voxels(grid) < (1 << 27)
                        and len(self.sobolev_kernel) in dev.LISTED_TAP_COUNTS):
                    boxes = dev.band_boxes(prepared, _lib.BAND_ALL)
                self._sobolev_boxes_used = boxes is not None
                sob = _SobolevStatePlan(f, states, canonical, grid, self.params, bands, g4, self.sobolev_kernel,
                                        self.min_iterations, n_max, gradient_every_iteration=every, boxes=boxes)
                self._sobolev_band = _Counted(sum(b.count for b in bands))  # what bench.py prices this path over
            if slab:
                self._plan_slab(f, live, grid, bands, 0 if self.min_iterations == 0
                                else max(self.max_iterations, self.min_iterations),
                                prepared if fused_prepare and prepared.cuts is not None else None)
        # with min_iterations == 0 the reference never enters its loop (max_warp starts at +inf, :354,:360-362)
        limit = 0 if self.min_iterations == 0 else max(self.max_iterations, self.min_iterations)
        it, n_exec = 0, 0
        dec = None
        early = None
        hooked = self.iteration_hook is not None
        while it < limit:
            # a run whose stop test cannot fire (min_iterations == max_iterations) has nothing to look at in between: all of
            # it is enqueued at once, whatever check_interval says
            batch = 1 if hooked else (limit - it if self.min_iterations >= limit else min(self.check_interval, limit - it))
            for i in range(it, it + batch):
                if planar_sobolev:
                    self._enqueue_iteration(i, lives[i % 2], lives[(i + 1) % 2], warps[i % 2], warps[(i + 1) % 2],
                                            canonical, grid, records, gbufs, limit)
                elif sob is not None:
                    sob.enqueue(i)
                else:
                    self._enqueue_state_iteration(i, states, limit)
            # z-slab: a gated run's device-side gate reads the records, so they are all-reduced (global max: idempotent;
            # once per record the energy sums of this batch); a fixed count only needs them on the host: every rank's
            # partial slots are gathered when they are read
            ungated = self.min_iterations >= limit
            if slab and not ungated:
                self.comm.reduce_records(records, it, it + batch)
            it += batch
            if finalize is not None and not planar_sobolev and it == limit and self.min_iterations >= limit:
                # every launch runs ungated, so the final state is states[limit % 2]: finalize before looking.  A slab run
                # may still have to be discarded (see optimize()), and the pass writes the caller's tensor: optimize() puts
                # the copy taken below back before it runs the call again
                early = SlavchevaOutcome(grid, canonical, state=states[limit % 2], listed=listed,
                                         sparse=prepared if sparse else None, warp_zeroed=warp_zeroed)
                if sparse:  # the pass must not touch the caller's fields when an update outran the initialised region
                    early.guard(records, limit, float(SPARSE_REACH))
                if slab and finalize[0] is not None:
                    self._slab_restore = (finalize[0], live_at_entry)
                early.enqueue_finalize(*finalize)
            dec = dev.decode_records(self.comm.gather_records(records, 0, it) if slab and ungated
                                     else dev.records_to_host(records[:it]))
            n_exec = int(dec["executed"].sum())
            # z-slab: every EXECUTED iteration must have stayed inside what the halo schedule keeps valid -- also those
            # of a batch in which the gate then closed (a large update followed by convergence inside one
            # check_interval).  Every rank sees the same reduced / gathered records, so the raise is collective.
            if sparse and not slab and n_exec > 0 and not dec["max_value"][:n_exec].max() < SPARSE_REACH:
                raise _SparseStateExceeded()  # (a slab's exchange groups stop at one voxel: _HaloTooNarrow below)
            reach = self.comm.layout.halo if slab else 0
            if slab and not self.sobolev and self._fast.exchange_interval > 1:
                reach = 1  # inside an exchange group every iteration may consume one slice of validity only
            if slab and n_exec > 0 and not (dec["max_value"][:n_exec].max() < reach):
                raise _HaloTooNarrow(float(dec["max_value"][:n_exec].max()), reach)
            if n_exec < it:
                break
            m = dec["max_value"][n_exec - 1]
            if hooked:
                self._call_hook(it - 1, float(m), lives, warps, states, canonical, grid, sob)
            if n_exec >= self.min_iterations and not (np.float32(self.lo) < m < np.float32(self.hi)):
                break
        self.iteration_count = n_exec
        wd, ws, wl = self.weights
        if dec is None:
            dec = dev.decode_records(dev.records_to_host(records[:1]))
        self.log = dict(max_warps=dec["max_value"][:n_exec].tolist(),
                        max_warp_indices=dec["argmax"][:n_exec].tolist(),
                        data_energies=(wd * dec["data_energy"][:n_exec]).tolist(),
                        smoothing_energies=(ws * dec["smoothing_energy"][:n_exec]).tolist(),
                        level_set_energies=(wl * dec["level_set_energy"][:n_exec]).tolist())
        if planar_sobolev:
            outcome = SlavchevaOutcome(grid, canonical, live=lives[n_exec % 2], warp_planar=warps[n_exec % 2])
        elif early is not None and n_exec == limit:
            outcome = early
        else:
            outcome = SlavchevaOutcome(grid, canonical, state=states[n_exec % 2], listed=listed,
                                       sparse=prepared if sparse else None, warp_zeroed=warp_zeroed)
        # what is needed to (re)produce gradient_field of the last executed iteration on demand
        if n_exec == 0:
            self._gradient_state = ("zeros", torch.zeros((dims,) + tuple(live.shape), dtype=torch.float32,
                                                         device=live.device))
        elif sob is not None:
            # the gradient buffers rotate identically every iteration, so the last executed iteration's filtered gradient is
            # in the buffer the (gated, skipped) later launches would have used too
            self._gradient_state = ("float4", sob, dims)
        elif self.sobolev:
            self._gradient_state = ("ready", self._last_g)
        else:
            self._gradient_state = ("recompute", states[(n_exec - 1) % 2], canonical, grid,
                                    (prepared, live) if sparse else None)
        return outcome

    def _optimize_run(self, live, canonical, grid, finalize):
        """_optimize for the case the library enqueues in one piece (lsf_state_run_begin / _finish): prepare pass, (sparse)
        states, band lists, all iterations, the listed finalize pass and the read-backs -- the same launches in the same
        order as the general path below makes one by one, hence the same results, in two foreign calls that run without the
        interpreter lock."""
        live_out, lower_threshold, statistics = finalize
        iterations = self.min_iterations
        device = live.device
        n = dev.n_voxels(grid)
        whole = dev.full_range(grid)
        sparse = (SPARSE_REACH > 0 and n >= SPARSE_MIN_VOXELS and not getattr(self, "_sparse_disabled", False))
        self._sparse_used = sparse
        usable = (live_out is not None and live_out.is_cuda and live_out.dtype == torch.float32
                  and live_out.is_contiguous() and tuple(live_out.shape) == tuple(live.shape))
        target = live_out if usable else torch.empty_like(live)
        if target is not live:
            target.copy_(live)  # the finalize pass writes listed voxels only
        states = [torch.empty(tuple(live.shape) + (4,), dtype=torch.float32, device=device) for _ in range(2)]
        scratch = torch.empty(int(_lib.lib.lsf_state_prepare_scratch_elements(ctypes.byref(whole))), dtype=torch.int32,
                              device=device)
        totals = torch.empty(5, dtype=torch.int64, device=device)
        totals_host = dev.pinned_scratch("run totals", 5, torch.int64)
        run = _lib.StateRun()
        run.live, run.canonical = dev._ptr(live, n, "live"), dev._ptr(canonical, n, "canonical")
        run.state[0], run.state[1] = states[0].data_ptr(), states[1].data_ptr()
        run.prepare_scratch, run.totals_device, run.totals_host = scratch.data_ptr(), totals.data_ptr(), totals_host.data_ptr()
        run.grid = whole
        run.sparse_reach = SPARSE_REACH if sparse else 0
        run.second_state_late = int(not sparse and n <= dev.StatePrepare.SPLIT_MAX_VOXELS)
        count_boxes = dev.boxes_ok(whole) and (self.box_walk is True or
                                               (self.box_walk is None and n >= BOX_WALK_MIN_VOXELS))
        box_scratch = None
        if count_boxes:
            box_scratch = torch.empty(int(_lib.lib.lsf_band_boxes_scratch_elements(ctypes.byref(whole))),
                                      dtype=torch.int32, device=device)
            run.box_scratch = box_scratch.data_ptr()
        stream = dev.stream_ptr()
        _lib.check(_lib.lib.lsf_state_run_begin(ctypes.byref(run), stream), "lsf_state_run_begin")
        # (the card is writing the states now; the lists are sized from the totals the call waited for)
        n_interior, n_boundary, opposite, first_opposite, n_boxes = totals_host.tolist()
        lists = torch.empty(max(n_interior + n_boundary, 1), dtype=torch.int32, device=device)
        boxes = box_canonical = None
        if n_boxes and (self.box_walk is True or 32 * n_interior > BOX_WALK_MIN_BAND_BYTES):
            boxes = torch.empty((n_boxes, 2), dtype=torch.int64, device=device)
            box_canonical = torch.empty(n_boxes * dev.BOX_EDGE ** 3, dtype=torch.float32, device=device)
        self._box_walk_used = boxes is not None
        records = dev.new_records(iterations, device)
        n_words = iterations * _lib.RECORD_SLOTS * dev.USED_SLOT_WORDS
        words = torch.empty(n_words + 16, dtype=torch.int64, device=device)  # the records' used words, then the statistics
        words_host = dev.pinned_scratch("run records", words.numel(), torch.int64)
        stats = stats_scratch = None
        if statistics:
            stats = torch.empty(16, dtype=torch.float64, device=device)
            stats_scratch = torch.empty(2 * int(_lib.lib.lsf_state_finalize_scratch_elements(ctypes.byref(whole))),
                                        dtype=torch.float64, device=device)
        max_value, argmax = np.empty(iterations, np.float32), np.empty(iterations, np.int64)
        energies, executed = np.empty((iterations, 3), np.float64), np.empty(iterations, np.bool_)
        result = _lib.StateRunResult(max_value.ctypes.data, argmax.ctypes.data, energies.ctypes.data, executed.ctypes.data)
        none = ctypes.c_void_p(0)
        p_lists = lists.data_ptr()
        # everything about the call's aftermath that does not depend on its results is made BEFORE the blocking call below --
        # behind it the card idles until the next call's first launch (tools/host_tail.py)
        bands = []
        if n_interior:
            bands.append(dev.BandList(lists[:n_interior], n_interior, _lib.BAND_INTERIOR))
        if n_boundary or not bands:
            bands.append(dev.BandList(lists[n_interior:] if n_boundary else lists[:1], n_boundary, _lib.BAND_BOUNDARY))
        f = _Counted(sum(b.count for b in bands))
        f.bands, f.records, f.boxes = bands, records, (boxes, box_canonical)
        outcome = _RunOutcome(grid, canonical, None, target, bands, None)
        weights = tuple(self.weights)
        _lib.check(_lib.lib.lsf_state_run_finish(
            ctypes.byref(run), ctypes.byref(self.params), ctypes.c_void_p(p_lists),
            ctypes.c_void_p(p_lists + 4 * n_interior), ctypes.c_void_p(boxes.data_ptr() if boxes is not None else 0),
            ctypes.c_void_p(box_canonical.data_ptr() if boxes is not None else 0),
            ctypes.c_void_p(records.data_ptr()), iterations,
            dev._ptr(target, n, "live_out"), float(lower_threshold),
            ctypes.c_void_p(stats.data_ptr()) if statistics else none,
            ctypes.c_void_p(stats_scratch.data_ptr()) if statistics else none, ctypes.c_void_p(words.data_ptr()),
            ctypes.c_void_p(words_host.data_ptr()), ctypes.byref(result), stream), "lsf_state_run_finish")
        if result.reach_exceeded:
            raise _SparseStateExceeded()  # the pass has left the caller's array alone (its guard); optimize() repeats
        n_exec = iterations if executed.all() else int(executed.sum())
        self._fast = f
        self.iteration_count = n_exec
        self.log = _RunLog(max_value[:n_exec], argmax[:n_exec], energies[:n_exec], weights)
        # gradient_field() recomputes the last iteration's gradient on demand from its INPUT state, at the listed voxels
        self._gradient_state = ("recompute_listed", states[(n_exec - 1) % 2], canonical, grid, bands)
        outcome.state = states[n_exec % 2]
        if statistics:
            outcome._raw = words_host[n_words:].numpy().view(np.float64).copy()
        return outcome

    def _call_hook(self, i, max_warp, lives, warps, states, canonical, grid, sob=None):
        """iteration i has run: hand its warp and gradient to the hook in the API layout (owned slices of a slab)"""
        if self.sobolev and sob is None:
            warp_planar, g = warps[(i + 1) % 2], self._last_g
        elif sob is not None:
            live_now = torch.empty(tuple(states[0].shape[:-1]), dtype=torch.float32, device=states[0].device)
            warp_planar = torch.empty((grid.dims,) + tuple(live_now.shape), dtype=torch.float32, device=live_now.device)
            dev.state_unpack(states[(i + 1) % 2], dev.full_range(grid), live_now, warp_planar, None)
            g = sob.final_gradient_planar(grid.dims)
        else:
            live_now = torch.empty(tuple(states[0].shape[:-1]), dtype=torch.float32, device=states[0].device)
            warp_planar = torch.empty((grid.dims,) + tuple(live_now.shape), dtype=torch.float32, device=live_now.device)
            dev.state_unpack(states[(i + 1) % 2], dev.full_range(grid), live_now, warp_planar, None)
            self._gradient_state = ("recompute", states[i % 2], canonical, grid, None)
            g = self.gradient_field()
        own = (slice(None), self.comm.layout.owned_local()) if self._slab() else (slice(None),)
        self.iteration_hook(0, i, dev.interleave(warp_planar[own].contiguous()), dev.interleave(g[own].contiguous()),
                            max_warp)

    def gradient_field(self):
        """planar gradient of the last executed iteration (zeroed where the live field snapped, DIRECT only).
        The fused kernel does not store it (12 B/voxel/iteration saved); it is recomputed here from the inputs
        of the last iteration, which the ping-pong buffers still hold, by the unfused kernels -- same code path,
        same bits."""
        st = self._gradient_state
        if st is None:
            return None
        if st[0] in ("zeros", "ready"):
            return st[1]
        if st[0] == "float4":  # SobolevFusion on the float4 layouts: the final gradient, made planar on demand
            g = st[1].final_gradient_planar(st[2])
            self._gradient_state = ("ready", g)
            return g
        if st[0] == "wide":  # the call was re-run on a wider internal slab: its gradient, cut to this slab's slices
            g = st[1].gradient_field()
            ax = st[3] if len(st) > 3 else 0
            return None if g is None else g.narrow(1 + ax, st[2].start, st[2].stop - st[2].start).contiguous()
        if st[0] == "recompute_listed":
            # the unfused kernels at the voxels of the call's band lists (the gradient is zero everywhere else): of the
            # input state only the listed voxels' neighbourhoods and re-warp cells are read, so a state that was initialised
            # near the band only (lsf_state_pack_needed) serves as it stands, and nothing of the caller's is touched
            _, state_in, canonical, grid, bands = st
            live_in = torch.empty(tuple(state_in.shape[:-1]), dtype=torch.float32, device=state_in.device)
            warp_in = torch.empty((grid.dims,) + tuple(live_in.shape), dtype=torch.float32, device=state_in.device)
            dev.state_unpack(state_in, grid, live_in, warp_in, None)
            g = torch.zeros_like(warp_in)
            scratch_records = dev.new_records(1, live_in.device)
            params = _lib.SlavchevaParams.from_buffer_copy(self.params)
            params.energy_mode = _lib.ENERGY_NONE
            warp_out, live_scratch = torch.empty_like(warp_in), torch.empty_like(live_in)
            for band in bands:
                if band.count:
                    dev.slavcheva_gradient(live_in, canonical, warp_in, g, grid, params, None, scratch_records, 0, band)
            for band in bands:
                if band.count:
                    dev.slavcheva_update_rewarp(live_in, canonical, g, warp_out, live_scratch, grid, params, None,
                                                scratch_records, 0, band)
            self._gradient_state = ("ready", g)
            return g
        _, state_in, canonical, grid = st[:4]
        if len(st) > 4 and st[4] is not None:
            # the state was initialised near the band only: complete it from the call's live array, which still holds the
            # input wherever no list entry points (the finalize pass writes listed voxels only)
            st[4][0].complete(state_in, st[4][1])
        live_in = torch.empty(tuple(state_in.shape[:-1]), dtype=torch.float32, device=state_in.device)
        warp_in = torch.empty((grid.dims,) + tuple(live_in.shape), dtype=torch.float32, device=state_in.device)
        dev.state_unpack(state_in, grid, live_in, warp_in, None)
        g = torch.empty_like(warp_in)
        scratch_records = dev.new_records(1, live_in.device)
        params = _lib.SlavchevaParams.from_buffer_copy(self.params)
        params.energy_mode = _lib.ENERGY_NONE
        dev.slavcheva_gradient(live_in, canonical, warp_in, g, grid, params, None, scratch_records, 0)
        dev.slavcheva_update_rewarp(live_in, canonical, g, torch.empty_like(warp_in), torch.empty_like(live_in), grid,
                                    params, None, scratch_records, 0)
        self._gradient_state = ("ready", g)
        return g
