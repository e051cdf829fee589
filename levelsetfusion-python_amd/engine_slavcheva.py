"""SlavchevaEngine: the per-iteration-update optimizer with in-place re-warping of the live field
(nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:332-408), D = 2 or 3, on whole volumes or slabs.  The drop-in classes
SlavchevaOptimizer2d / 3d are thin shells around it.  This module: construction, the general optimize() path (one foreign
call per launch, batches of `check_interval` iterations between host reads), hooks and gradient_field; the
library-enqueued calls live in engine_run.py, slabs in engine_slab.py, the SobolevFusion launch plans in
engine_sobolev.py, what a call leaves behind in engine_outcome.py."""
import ctypes

import numpy as np
import torch

from . import _lib, device as dev, engine_options
from .engine_common import _Counted
from .engine_focus import FocusMixin
from .engine_outcome import SlavchevaOutcome
from .engine_run import RunMixin, _SparseStateExceeded
from .engine_slab import SlabMixin, _HaloTooNarrow
from .engine_sobolev import PlanarSobolevMixin, _SobolevStatePlan


class SlavchevaEngine(RunMixin, SlabMixin, PlanarSobolevMixin, FocusMixin):
    """per-iteration-update optimizer with in-place re-warping of the live field
    (nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:332-408), D = 2 or 3, optionally on a z-slab."""

    def __init__(self, direct, level_set_term_enabled, sobolev_smoothing_enabled, data_term_method,
                 smoothing_term_method, gradient_descent_rate, data_term_weight, smoothing_term_weight,
                 isomorphic_enforcement_factor, level_set_term_weight, lower_threshold, upper_threshold,
                 max_iterations, min_iterations, sobolev_kernel, compute_energies=True, check_interval=32,
                 comm=None, options=None):
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
        engine_options.apply(self, engine_options.SLAVCHEVA_DEFAULTS, options)  # use_band_list, library_run, ...
        self.last_call = engine_options.new_call_report()
        # a call on sparsely initialised states met an update they cannot carry: this optimizer's data move that far,
        # every later call runs on fully initialised states
        self.sparse_disabled = False
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
        return self._enqueue_slab_state_iteration(i, states, limit)

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
                self.sparse_disabled = True
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
        self.last_call = engine_options.new_call_report()
        grid = self._grid(live)
        dims = grid.dims
        n_rec = max(self.max_iterations, self.min_iterations, 1)
        slab = self._slab()
        # somebody looks at every iteration (the hook, the focus-neighbourhood trace): launched and synchronised one by one
        watched = self.iteration_hook is not None or self.focus_voxels is not None
        if self.focus_voxels is not None:
            self._focus_begin(live, canonical)
        run_ok = (self.library_run and finalize is not None and not self.sobolev and self.use_band_list
                  and not watched and self.min_iterations > 0 and dev.buffer_addressing_ok(grid))
        if (self.library_run and finalize is not None and self.sobolev and self.sobolev_boxes and self.use_band_list
                and not watched and self.min_iterations > 0 and not slab and grid.dims == 3
                and dev.boxes_ok(grid) and dev.n_voxels(grid) < (1 << 27) and dev.buffer_addressing_ok(grid)
                and len(self.sobolev_kernel) in dev.LISTED_TAP_COUNTS):
            # SobolevFusion on a whole 3-D volume of whole boxes: the call enqueued by the library as well
            return self._optimize_run(live, canonical, grid, finalize, sobolev=True)
        if run_ok and not slab:
            # a whole volume, no Sobolev filter, nobody watching the iterations: the whole call is enqueued by the library
            # (two host calls; slavcheva_optimizer2d.py:354-388's loop without a Python iteration) -- a fixed iteration count
            # at once, the reference's default threshold-terminated loop in batches of check_interval gated launches
            return self._optimize_run(live, canonical, grid, finalize)
        if run_ok and slab and not finalize[2] and self.min_iterations >= self.max_iterations and self._slab_run_ok(grid):
            # a z-slab rank on the library's RCCL transport, a fixed iteration count: likewise (lsf_slab_run_begin / _finish)
            return self._optimize_slab_run(live, canonical, grid, finalize)
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
                           and self.comm.layout.halo >= max(self.sparse_reach, 2))
            sparse = ((not slab or slab_groups) and self.sparse_reach > 0 and dev.n_voxels(grid) >= self.sparse_min_voxels
                      and not watched and not self.sparse_disabled)
            # the listed finalize pass wants a zero-filled warp output (192 MB at 256^3, 26 us): filled in the call's
            # prologue, where the card waits for the host, instead of behind the last iteration.  Up to 256^3 IN FRONT of
            # the counting pass: the states written behind it are then the last thing to pass through the 256 MB Infinity
            # Cache before the first two iterations read them (filled behind the states it evicted them: 1.846-1.858
            # against 1.817-1.825 ms per step, three alternating runs on one box); a larger volume's fill (1.6 GB at
            # 512^3) would only keep the list sizes from the host (8.67 against 8.53 ms)
            fill_first = dev.n_voxels(grid) <= (1 << 24)
            if finalize is not None and not slab and fill_first:
                warp_zeroed = torch.zeros(tuple(live.shape) + (dims,), dtype=torch.float32, device=live.device)
            self.last_call.sparse_states = sparse
            prepared = dev.StatePrepare(live, canonical, dev.full_range(grid), cut_chunks,
                                        sparse_reach=self.sparse_reach if sparse else 0)
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
                self.last_call.sobolev_boxes = boxes is not None
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
        hooked = watched
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
                    early.guard(records, limit, float(self.sparse_reach))
                if slab and finalize[0] is not None:
                    self._slab_restore = (finalize[0], live_at_entry)
                early.enqueue_finalize(*finalize)
            dec = dev.decode_records(self.comm.gather_records(records, 0, it) if slab and ungated
                                     else dev.records_to_host(records[:it]))
            n_exec = int(dec["executed"].sum())
            # z-slab: every EXECUTED iteration must have stayed inside what the halo schedule keeps valid -- also those
            # of a batch in which the gate then closed (a large update followed by convergence inside one
            # check_interval).  Every rank sees the same reduced / gathered records, so the raise is collective.
            if sparse and not slab and n_exec > 0 and not dec["max_value"][:n_exec].max() < self.sparse_reach:
                raise _SparseStateExceeded()  # (a slab's exchange groups stop at one voxel: _HaloTooNarrow below)
            reach = self.comm.layout.halo if slab else 0
            if slab and not self.sobolev and self._fast.exchange_interval > 1:
                reach = 1  # inside an exchange group every iteration may consume one slice of validity only
            if slab and n_exec > 0 and not (dec["max_value"][:n_exec].max() < reach):
                raise _HaloTooNarrow(float(dec["max_value"][:n_exec].max()), reach)
            if n_exec < it:
                break
            m = dec["max_value"][n_exec - 1]
            if self.focus_voxels is not None:
                self._probe_focus(it - 1, lives, warps, states, canonical, grid)
            if self.iteration_hook is not None:
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
