"""Whole optimize() calls enqueued by the LIBRARY (csrc/lsf_slavcheva_run.hip): the loop of
slavcheva_optimizer2d.py:354-388 without a Python iteration -- two foreign calls that run without the interpreter lock."""
import ctypes

import numpy as np
import torch

from . import _lib, device as dev
from .engine_common import _Counted
from .engine_outcome import _RunLog, _RunOutcome


class _SparseStateExceeded(Exception):
    """a call whose ping-pong states were initialised near the band only (dev.StatePrepare(sparse_reach=...)) met a warp
    update that can read beyond that; nothing of the caller's has been modified (SlavchevaEngine.optimize repeats the
    call on fully initialised states and keeps doing so for this optimizer)"""


class _RunGradient:
    """the SobolevFusion call's final gradient (float4 [z,]y,x,4), handed out planar like _SobolevStatePlan's"""

    def __init__(self, g4):
        self._g4 = g4

    def final_gradient_planar(self, dims):
        return self._g4[..., :dims].movedim(-1, 0).contiguous()


class RunMixin:
    """SlavchevaEngine's library-enqueued calls on whole volumes"""

    def _optimize_run(self, live, canonical, grid, finalize, sobolev=False):
        """_optimize for the case the library enqueues in one piece (lsf_state_run_begin / _finish): prepare pass, (sparse)
        states, band lists, all iterations -- a fixed count, or the reference's threshold-terminated loop in batches of
        check_interval gated launches --, the listed finalize pass and the read-backs: the same launches in the same order
        as the general path makes one by one, hence the same results, in two foreign calls that run without the interpreter
        lock.  sobolev: the SobolevFusion iteration on whole 3-D volumes of whole boxes (lsf_sobolev_run_finish: gradient + x
        pass over the list of all band voxels, then y pass, z pass, update and re-warp box by box)."""
        live_out, lower_threshold, statistics = finalize
        # the number of records; with min < max the stop test can fire (the reference's default: slavcheva_optimizer2d.py:
        # 360-362) and the library enqueues check_interval gated iterations at a time, reading the records in between
        iterations = max(self.min_iterations, self.max_iterations)
        loop = None
        if self.min_iterations < iterations:
            loop = _lib.RunLoop(self.min_iterations, self.max_iterations, self.lo, self.hi, self.check_interval, 0)
        device = live.device
        n = dev.n_voxels(grid)
        whole = dev.full_range(grid)
        sparse = (self.sparse_reach > 0 and n >= self.sparse_min_voxels and not self.sparse_disabled)
        self.last_call.sparse_states = sparse
        self.last_call.library_run = True
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
        run.sparse_reach = self.sparse_reach if sparse else 0
        run.second_state_late = int(not sparse and n <= dev.StatePrepare.SPLIT_MAX_VOXELS)
        count_boxes = sobolev or (dev.boxes_ok(whole) and (self.box_walk is True or
                                                           (self.box_walk is None and n >= self.box_walk_min_voxels)))
        run.box_all = int(sobolev)
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
        if sobolev:
            boxes = torch.empty((max(n_boxes, 1), 2), dtype=torch.int64, device=device)
            list_all = torch.empty(max(n_interior + n_boundary, 1), dtype=torch.int32, device=device) \
                if n_interior and n_boundary else None
            g4 = self._sobolev_gradient_buffers(live, whole, lists, n_interior + n_boundary)
            taps = np.ascontiguousarray(np.asarray(self.sobolev_kernel, dtype=np.float64))
            self.last_call.sobolev_boxes = True
        elif n_boxes and (self.box_walk is True or 32 * n_interior > self.box_walk_min_band_bytes):
            boxes = torch.empty((n_boxes, 2), dtype=torch.int64, device=device)
            box_canonical = torch.empty(n_boxes * dev.BOX_EDGE ** 3, dtype=torch.float32, device=device)
        self.last_call.box_walk = boxes is not None and not sobolev
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
        if sobolev:
            f.keep = (g4, list_all, taps)
            self._sobolev_band = f  # what bench.py prices this path over
            _lib.check(_lib.lib.lsf_sobolev_run_finish(
                ctypes.byref(run), ctypes.byref(self.params), taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                int(taps.size), ctypes.c_void_p(p_lists), ctypes.c_void_p(p_lists + 4 * n_interior),
                ctypes.c_void_p(list_all.data_ptr() if list_all is not None else 0), ctypes.c_void_p(boxes.data_ptr()),
                ctypes.c_void_p(g4[0].data_ptr()), ctypes.c_void_p(g4[1].data_ptr()), ctypes.c_void_p(records.data_ptr()),
                iterations, ctypes.byref(loop) if loop is not None else None, dev._ptr(target, n, "live_out"),
                float(lower_threshold), ctypes.c_void_p(stats.data_ptr()) if statistics else none,
                ctypes.c_void_p(stats_scratch.data_ptr()) if statistics else none, ctypes.c_void_p(words.data_ptr()),
                ctypes.c_void_p(words_host.data_ptr()), ctypes.byref(result), stream), "lsf_sobolev_run_finish")
        else:
            self._run_finish(run, p_lists, n_interior, boxes, box_canonical, records, iterations, loop, target, n,
                             lower_threshold, stats, stats_scratch, statistics, words, words_host, result, stream)
        return self._after_run(result, executed, iterations, f, max_value, argmax, energies, weights, states, canonical, grid,
                               bands, outcome, statistics, words_host, n_words, g4[1] if sobolev else None)

    def _sobolev_gradient_buffers(self, live, whole, lists, n_listed):
        """the two float4 gradient buffers of a library-enqueued SobolevFusion call, all zero: the previous call's buffers with
        zeros written back at ITS listed voxels (every other voxel was never written: lsf_zero_listed4, 2 x 26 MB at 256^3) or,
        for another shape, fresh zero fills (2 x 268 MB).  The optimizer keeps them between calls (the reference allocates its
        gradient per call, slavcheva_optimizer2d.py:343-346)."""
        key = (tuple(live.shape), live.device)
        cache = getattr(self, "_g4_cache", None)
        if cache is not None and cache[0] == key:
            _, g4, old_lists, old_n = cache
            for t, bricks in ((g4[0], 1), (g4[1], 0)):
                _lib.check(_lib.lib.lsf_zero_listed4(ctypes.c_void_p(t.data_ptr()), ctypes.byref(whole),
                                                     ctypes.c_void_p(old_lists.data_ptr()), old_n, bricks, dev.stream_ptr()),
                           "lsf_zero_listed4")
        else:
            self._g4_cache = None  # (let go of another shape's buffers first)
            g4 = [torch.zeros(tuple(live.shape) + (4,), dtype=torch.float32, device=live.device) for _ in range(2)]
        # recorded BEFORE the call runs: whatever happens in it, these are the voxels it may have written
        self._g4_cache = (key, g4, lists, int(n_listed))
        return g4

    def _run_finish(self, run, p_lists, n_interior, boxes, box_canonical, records, iterations, loop, target, n,
                    lower_threshold, stats, stats_scratch, statistics, words, words_host, result, stream):
        none = ctypes.c_void_p(0)
        _lib.check(_lib.lib.lsf_state_run_finish(
            ctypes.byref(run), ctypes.byref(self.params), ctypes.c_void_p(p_lists),
            ctypes.c_void_p(p_lists + 4 * n_interior), ctypes.c_void_p(boxes.data_ptr() if boxes is not None else 0),
            ctypes.c_void_p(box_canonical.data_ptr() if boxes is not None else 0),
            ctypes.c_void_p(records.data_ptr()), iterations, ctypes.byref(loop) if loop is not None else None,
            dev._ptr(target, n, "live_out"), float(lower_threshold),
            ctypes.c_void_p(stats.data_ptr()) if statistics else none,
            ctypes.c_void_p(stats_scratch.data_ptr()) if statistics else none, ctypes.c_void_p(words.data_ptr()),
            ctypes.c_void_p(words_host.data_ptr()), ctypes.byref(result), stream), "lsf_state_run_finish")

    def _after_run(self, result, executed, iterations, f, max_value, argmax, energies, weights, states, canonical, grid, bands,
                   outcome, statistics, words_host, n_words, sobolev_gradient):
        if result.reach_exceeded:
            raise _SparseStateExceeded()  # the pass has left the caller's array alone (its guard); optimize() repeats
        n_exec = iterations if executed.all() else int(executed.sum())
        self._fast = f
        self.iteration_count = n_exec
        self.log = _RunLog(max_value[:n_exec], argmax[:n_exec], energies[:n_exec], weights)
        if sobolev_gradient is not None:
            # the filtered gradient of the last executed iteration (float4, made planar when read)
            self._gradient_state = ("float4", _RunGradient(sobolev_gradient), grid.dims)
        else:
            # gradient_field() recomputes the last iteration's gradient on demand from its INPUT state, at the listed voxels
            self._gradient_state = ("recompute_listed", states[(n_exec - 1) % 2], canonical, grid, bands)
        outcome.state = states[n_exec % 2]
        if statistics:
            outcome._raw = words_host[n_words:].numpy().view(np.float64).copy()
        return outcome
