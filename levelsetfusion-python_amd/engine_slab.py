"""SlavchevaEngine on z- / y-slabs (one process per GPU): the launch plan of a slab rank --
exchange groups, boundary-first split, deferred waits, compact faces --, the per-iteration enqueue through the
library's RCCL transport or torch.distributed, and the wider re-run of a call whose updates outgrow the halo schedule.
The reference has no distributed code; its caller shape is the loop over independent pairs of
run_hierarchical_optimizer3d_multipair.py:403-432."""
import ctypes
import os

import numpy as np
import torch

from . import _lib, device as dev
from .engine_common import _Counted, _Lazy
from .engine_outcome import SlavchevaOutcome, _RunLog, _RunOutcome


class _HaloTooNarrow(Exception):
    """a z-slab run met a warp update its halo schedule cannot carry (SlavchevaEngine.optimize re-runs it wider)"""

    def __init__(self, max_update, validity):
        super().__init__("warp update of %.3f voxels against %d slice(s) of validity" % (max_update, validity))
        self.max_update = float(max_update)



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


class SlabMixin:
    """the slab half of SlavchevaEngine (fused path on the float4 state)"""

    def _enqueue_slab_state_iteration(self, i, states, limit):
        """one iteration of a slab rank: what it launches and whether the faces travel afterwards is planned in _plan_slab"""
        f = self._fast
        s_in, s_out = f.p_state[i % 2], f.p_state[(i + 1) % 2]
        gate_ref = None if i < self.min_iterations else f.gate_ref(i - 1)
        run = _lib.lib.lsf_slavcheva_state_iteration
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
                return _Parts([])
            if listed:  # one launch per subset over all ranges
                return _Parts([(grid_of(ranges[0][0], ranges[-1][1]), lists_of(ranges))])
            return _Parts([(grid_of(z0, z1), [dev.BandList.none()]) for z0, z1 in ranges])

        # Every (boundary part, interior part) pair is built when an iteration first asks for it: only the first
        # iteration's pair stands between the list sizes and the first launch, the others are made while launches are
        # already queued (13 descriptors, ~0.1 ms of host work at 256^3)
        empty = _Parts([])
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
                return _Parts([])
            return _Parts([(grid, lists_of(ranges))])  # the grid already carries the owned rows (energies)

        empty = _Parts([])
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
        # every part list is a filtered copy (boolean masks: device -> host reads of the counts): made NOW, while the card is
        # still busy with the states' initialisation, instead of one by one between the iterations' launches, where each
        # read would drain the launch stream
        for lazy in f.widened_parts + [f.exchange_parts, f.resume_parts]:
            lazy.get()
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

    # ---- the whole call of a z-slab rank enqueued by the library (csrc/lsf_slab.hip: lsf_slab_run_begin / _finish) ---------
    def _slab_run_ok(self, grid):
        """can this slab call be handed to the library in one piece?  z-slabs on the native RCCL transport whose slices are a
        multiple of 1024 voxels (the cut positions of the schedule are per-chunk prefix counts of the counting pass)"""
        L = self.comm.layout
        return (L.axis == 0 and (grid.ny * grid.nx) % dev.StatePrepare.CHUNK == 0 and L.halo <= 32
                and L.z_end - L.z_begin >= 2 * L.halo and grid.nz <= 4095 and self.comm.native() is not None)

    def _optimize_slab_run(self, live, canonical, grid, finalize):
        """_optimize for a fixed-count call of a z-slab rank, enqueued by the library in two foreign calls: the counting pass
        with the schedule's cut positions, (sparse) states, list fills, the exchange groups of _plan_slab (boundary first,
        compact faces cross-checked with the neighbours, deferred waits), the listed finalize pass and ONE gather of every
        rank's records.  The same launches as _enqueue_slab_state_iteration makes one by one from Python
        (tests/slab_loopback_worker.py holds the two against each other bit for bit), without ~50 Python -> C calls, the
        part descriptors and two torch.distributed collectives per call."""
        live_out, lower_threshold, statistics = finalize
        L = self.comm.layout
        iterations = self.min_iterations
        device = live.device
        n = dev.n_voxels(grid)
        whole = dev.full_range(grid)  # the local array: owned slices + halos (z_global_offset stays)
        every = getattr(self, "_exchange_every_iteration", False)
        # sparse states in exchange groups only, with a halo of at least the reach (see _optimize)
        sparse = (not every and L.halo >= max(self.sparse_reach, 2) and self.sparse_reach > 0
                  and n >= self.sparse_min_voxels and not self.sparse_disabled)
        self.last_call.sparse_states = sparse
        self.last_call.library_run = True
        usable = (live_out is not None and live_out.is_cuda and live_out.dtype == torch.float32
                  and live_out.is_contiguous() and tuple(live_out.shape) == tuple(live.shape))
        target = live_out if usable else torch.empty_like(live)
        if target is not live:
            target.copy_(live)  # the finalize pass writes listed voxels only
        elif usable:
            # the pass writes the caller's tensor before every rank's records have said whether the call stands: what it
            # overwrites is kept (optimize() puts it back before a wider re-run)
            self._slab_restore = (live_out, live_out.clone())
        states = [torch.empty(tuple(live.shape) + (4,), dtype=torch.float32, device=device) for _ in range(2)]
        scratch = torch.empty(int(_lib.lib.lsf_state_prepare_scratch_elements(ctypes.byref(whole))), dtype=torch.int32,
                              device=device)
        n_totals = 5 + 2 * _lib.SLAB_MAX_CUTS
        totals = torch.empty(n_totals, dtype=torch.int64, device=device)
        totals_host = dev.pinned_scratch("slab run totals", n_totals, torch.int64)
        _, world, lo_rank, hi_rank = self.comm.native_identity()
        run = _lib.SlabRun()
        base = run.base
        base.live, base.canonical = dev._ptr(live, n, "live"), dev._ptr(canonical, n, "canonical")
        base.state[0], base.state[1] = states[0].data_ptr(), states[1].data_ptr()
        base.prepare_scratch, base.totals_device, base.totals_host = scratch.data_ptr(), totals.data_ptr(), \
            totals_host.data_ptr()
        base.grid = whole
        base.sparse_reach = self.sparse_reach if sparse else 0
        base.second_state_late = int(not sparse and n <= dev.StatePrepare.SPLIT_MAX_VOXELS)
        run.layout = _lib.SlabLayoutC(grid.nz, grid.ny, grid.nx, L.z_begin, L.z_end, L.halo, lo_rank, hi_rank)
        run.exchange_interval = 1 if every else L.halo
        stream = dev.stream_ptr()
        native = self.comm.native()
        _lib.check(_lib.lib.lsf_slab_run_begin(ctypes.byref(run), stream), "lsf_slab_run_begin")
        n_interior, n_boundary = totals_host[:2].tolist()
        lists = torch.empty(max(n_interior + n_boundary, 1), dtype=torch.int32, device=device)
        index_scratch = torch.empty(max(int(run.out_index_entries), 1), dtype=torch.int32, device=device)
        messages = None
        if os.environ.get("LSF_SLAB_FACES", "compact") != "full":
            messages = torch.empty(4 * int(run.out_face_entries) + 16, dtype=torch.float32, device=device)
        records = dev.new_records(iterations, device)
        n_words = iterations * _lib.RECORD_SLOTS * dev.USED_SLOT_WORDS
        words = torch.empty((1 + world) * n_words, dtype=torch.int64, device=device)
        words_host = dev.pinned_scratch("slab run records", world * n_words, torch.int64)
        max_value, argmax = np.empty(iterations, np.float32), np.empty(iterations, np.int64)
        energies, executed = np.empty((iterations, 3), np.float64), np.empty(iterations, np.bool_)
        result = _lib.StateRunResult(max_value.ctypes.data, argmax.ctypes.data, energies.ctypes.data, executed.ctypes.data)
        p_lists = lists.data_ptr()
        bands = []
        if n_interior:
            bands.append(dev.BandList(lists[:n_interior], n_interior, _lib.BAND_INTERIOR))
        if n_boundary or not bands:
            bands.append(dev.BandList(lists[n_interior:] if n_boundary else lists[:1], n_boundary, _lib.BAND_BOUNDARY))
        f = _Counted(sum(b.count for b in bands))
        f.bands, f.records, f.exchange_interval, f.keep = bands, records, int(run.exchange_interval), (index_scratch, messages)
        outcome = _RunOutcome(grid, canonical, None, target, bands, None)
        weights = tuple(self.weights)
        _lib.check(_lib.lib.lsf_slab_run_finish(
            ctypes.byref(run), native, ctypes.byref(self.params), ctypes.c_void_p(p_lists),
            ctypes.c_void_p(p_lists + 4 * n_interior), ctypes.c_void_p(index_scratch.data_ptr()),
            ctypes.c_void_p(messages.data_ptr() if messages is not None else 0), ctypes.c_void_p(records.data_ptr()),
            iterations, dev._ptr(target, n, "live_out"), ctypes.c_void_p(words.data_ptr()),
            ctypes.c_void_p(words_host.data_ptr()), ctypes.byref(result), stream), "lsf_slab_run_finish")
        f.compact_faces = int(result.compact_faces)
        if result.compact_faces == 0 and messages is not None:
            import warnings
            warnings.warn("slab halos are not consistent with the neighbours' slabs (band voxel counts differ): whole "
                          "slices are exchanged")
        self._fast = f
        n_exec = iterations if executed.all() else int(executed.sum())
        # every EXECUTED iteration must have stayed inside what the halo schedule keeps valid: one slice inside an exchange
        # group, the halo width with an exchange per iteration.  Every rank decoded the same gathered records, so the raise
        # is collective; optimize() restores the caller's tensor and runs the call again on a wider slab
        reach = 1 if run.exchange_interval > 1 else L.halo
        if n_exec > 0 and not (max_value[:n_exec].max() < reach):
            raise _HaloTooNarrow(float(max_value[:n_exec].max()), reach)
        self.iteration_count = n_exec
        self.log = _RunLog(max_value[:n_exec], argmax[:n_exec], energies[:n_exec], weights)
        self._gradient_state = ("recompute_listed", states[(n_exec - 1) % 2], canonical, whole, bands)
        outcome.state = states[n_exec % 2]
        return outcome

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
        self.iteration_count, self.log, self.last_call = clone.iteration_count, clone.log, clone.last_call
        off = L2.halo_lo - L.halo_lo
        window = slice(off, off + L.nz_local)
        self._gradient_state = ("wide", clone, window, ax)
        grid = self._grid(live)
        if outcome.state is not None:
            return SlavchevaOutcome(grid, canonical, state=outcome.state.narrow(ax, off, L.n_local).contiguous())
        return SlavchevaOutcome(grid, canonical, live=outcome.live().narrow(ax, off, L.n_local).contiguous(),
                                warp_planar=outcome.warp_planar().narrow(1 + ax, off, L.n_local).contiguous())
