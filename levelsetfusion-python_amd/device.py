"""Torch-facing wrappers around the C ABI of liblsf_hip.so.

torch is plumbing here: it owns device memory and the HIP stream.  Every wrapper validates dtype, device,
contiguity and element counts on the host BEFORE the launch (a hand-written kernel that reads past a buffer can
take the whole GPU down), then passes raw device pointers + the current HIP stream to the library.
The common helpers live in device_core.py, the field operations (warps, pyramids, filters) in device_fields.py; this module
re-exports both and holds the optimizers' kernels.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import Gate, Grid, HierParams, SlavchevaParams, check, lib
from .device_core import *  # noqa: F401,F403
from .device_core import _PINNED, _gate_ref, _ptr, _record_ptr  # noqa: F401
from .device_fields import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------- optimizer kernels
def hier_iteration(packed, canonical, warp, g_prev, g_out, grid, params, gate, records, index):
    n = n_voxels(grid)
    # params.packed_nz > 0: the gather operand holds that many slices (a wider or replicated copy of the level)
    n_packed = params.packed_nz * grid.ny * grid.nx if params.packed_nz > 0 else n
    check(lib.lsf_hier_iteration(_ptr(packed, 4 * n_packed, "packed live"), _ptr(canonical, n, "canonical"),
                                 _ptr(warp, n * grid.dims, "warp"),
                                 _ptr(g_prev, n * grid.dims, "g_prev", allow_none=True),
                                 _ptr(g_out, n * grid.dims, "g_out", allow_none=True), ctypes.byref(grid),
                                 ctypes.byref(params), _gate_ref(gate), _record_ptr(records, index), stream_ptr()),
          "lsf_hier_iteration")


def hier_update(g, warp, grid, rate, gate, records, index):
    """warp -= rate * g and record[index].max = max |g|; warp None: the maximum only (convolve_xyz moved the warp)"""
    n = n_voxels(grid) * grid.dims
    check(lib.lsf_hier_update(_ptr(g, n, "g"), _ptr(warp, n, "warp", allow_none=True), ctypes.byref(grid), float(rate),
                              _gate_ref(gate), _record_ptr(records, index), stream_ptr()), "lsf_hier_update")


class BandList:
    """ascending voxel indices of (a subset of) the narrow-band union inside the grid's z-range
    (lsf_band_list_fill); subset = _lib.BAND_ALL / BAND_INTERIOR / BAND_BOUNDARY"""

    def __init__(self, indices, count, subset=_lib.BAND_ALL):
        self.indices, self.count, self.subset = indices, int(count), int(subset)
        self.pointer = ctypes.c_void_p(indices.data_ptr() if indices is not None else 0)

    @classmethod
    def none(cls):
        """no list: the kernel walks every voxel"""
        return cls(None, 0)


def buffer_addressing_ok(grid, bytes_per_voxel=16):
    """can the list kernels take their buffer-load neighbourhood path?  Planar fields (4 B per voxel, the Sobolev path):
    32-bit offsets over all planes.  The float4 state (16 B): offsets are relative to a wave's first voxel
    (lsf_slavcheva_state_iteration), so only two slices + two rows have to fit 32 bits, and voxel indices int32."""
    if bytes_per_voxel != 16:
        return bytes_per_voxel * n_voxels(grid) < 0xffffffff
    return n_voxels(grid) < 0x7fffffff and 16 * (2 * grid.nx * grid.ny + 2 * grid.nx + 3) < 0x7fffffff


def band_lists(live, canonical, grid=None, split=True, bytes_per_voxel=16):
    """the band list(s) one fused iteration launches over, built with ONE host read of the totals: with `split` (and a
    field small enough for 32-bit buffer offsets) the INTERIOR voxels -- served by the kernel without out-of-bounds
    handling -- and the BOUNDARY voxels, empty lists dropped (but never both); otherwise one list of ALL band voxels"""
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    subsets = (_lib.BAND_INTERIOR, _lib.BAND_BOUNDARY) if split and buffer_addressing_ok(grid, bytes_per_voxel) \
        else (_lib.BAND_ALL,)
    n_scratch = int(lib.lsf_band_scratch_elements(ctypes.byref(grid)))
    scratch = torch.empty((len(subsets), n_scratch), dtype=torch.int32, device=live.device)
    totals = torch.zeros(len(subsets), dtype=torch.int64, device=live.device)
    p_live, p_canon = _ptr(live, n, "live"), _ptr(canonical, n, "canonical")
    for k, subset in enumerate(subsets):
        check(lib.lsf_band_count(p_live, p_canon, ctypes.byref(grid), subset, ctypes.c_void_p(scratch[k].data_ptr()),
                                 ctypes.c_void_p(totals[k:].data_ptr()), stream_ptr()), "lsf_band_count")
    counts = [int(c) for c in totals.cpu()]
    lists = []
    for k, (subset, count) in enumerate(zip(subsets, counts)):
        if count == 0 and not (k == len(subsets) - 1 and not lists):
            continue
        indices = torch.empty(max(count, 1), dtype=torch.int32, device=live.device)
        if count:
            check(lib.lsf_band_list_fill(p_live, p_canon, ctypes.byref(grid), subset,
                                         ctypes.c_void_p(scratch[k].data_ptr()), ctypes.c_void_p(indices.data_ptr()),
                                         stream_ptr()), "lsf_band_list_fill")
        lists.append(BandList(indices, count, subset))
    return lists


def band_list(live, canonical, grid=None, subset=_lib.BAND_ALL):
    """one band list of the given subset"""
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    scratch = torch.empty(int(lib.lsf_band_scratch_elements(ctypes.byref(grid))), dtype=torch.int32,
                          device=live.device)
    total = torch.zeros(1, dtype=torch.int64, device=live.device)
    p_live, p_canon = _ptr(live, n, "live"), _ptr(canonical, n, "canonical")
    check(lib.lsf_band_count(p_live, p_canon, ctypes.byref(grid), int(subset), ctypes.c_void_p(scratch.data_ptr()),
                             ctypes.c_void_p(total.data_ptr()), stream_ptr()), "lsf_band_count")
    count = int(total.item())
    indices = torch.empty(max(count, 1), dtype=torch.int32, device=live.device)
    if count:
        check(lib.lsf_band_list_fill(p_live, p_canon, ctypes.byref(grid), int(subset),
                                     ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(indices.data_ptr()),
                                     stream_ptr()), "lsf_band_list_fill")
    return BandList(indices, count, subset)


def slavcheva_gradient(live, canonical, warp_prev, g_out, grid, params, gate, records, index, band=None):
    """gradient + energies of one iteration on planar fields (the Sobolev path; lsf_slavcheva_gradient); band: an
    LSF_BAND_ALL list -- only its voxels are visited (g_out must hold zeros elsewhere)"""
    n = n_voxels(grid)
    nd = n * grid.dims
    check(lib.lsf_slavcheva_gradient(_ptr(live, n, "live"), _ptr(canonical, n, "canonical"),
                                     _ptr(warp_prev, nd, "warp_prev"), _ptr(g_out, nd, "g_out"), ctypes.byref(grid),
                                     ctypes.byref(params), _gate_ref(gate), _record_ptr(records, index),
                                     band.pointer if band is not None else ctypes.c_void_p(0),
                                     band.count if band is not None else 0, stream_ptr()),
          "lsf_slavcheva_gradient")


class StatePrepare:
    """start of a fused optimize() call on whole arrays, one pass over live and canonical (lsf_state_prepare): the two
    ping-pong states (live, 0) -- `states`, valid in stream order as soon as this object exists -- and, from collect(),
    the INTERIOR + BOUNDARY band lists, filled from the ballots the pass keeps (lsf_band_list_fill_prepared).  The
    constructor only launches: whatever the host has to set up for the iterations fits between it and collect(), which
    is where the host waits for the list sizes."""

    CHUNK = 1024  # voxels per count of lsf_state_prepare's scratch (kBandChunk)
    SPLIT_MAX_VOXELS = 1 << 24  # up to here the second state's fill fits into the host's wait for the list sizes

    def __init__(self, live, canonical, grid=None, cut_chunks=None, sparse_reach=0):
        """cut_chunks (optional): int64 device tensor of chunk indices <= the number of chunks -- the number of INTERIOR /
        BOUNDARY list entries in front of voxel 1024 * chunk comes back with the list sizes (collect), e.g. the positions
        of z cuts in the lists of a slab whose slices are a multiple of 1024 voxels, without a search and a second host
        read.  (The entry AT the number of chunks is not a count: use cut_totals for a cut at the end of the array.)
        sparse_reach > 0: the states are written ONLY in the 1024-voxel chunks an iteration can read while every update
        stays below `sparse_reach` voxels (lsf_state_pack_needed) -- the rest of both buffers stays uninitialised;
        complete(state, live) fills it in for whole-state readers."""
        self.grid = grid = grid or make_grid(live.shape)
        n = n_voxels(grid)
        self.states = [torch.empty(tuple(live.shape) + (4,), dtype=torch.float32, device=live.device) for _ in range(2)]
        n_scratch = int(lib.lsf_state_prepare_scratch_elements(ctypes.byref(grid)))
        self._scratch = torch.empty(n_scratch, dtype=torch.int32, device=live.device)
        totals = torch.empty(4, dtype=torch.int64, device=live.device)
        self.sparse_reach = int(sparse_reach)
        split = n <= self.SPLIT_MAX_VOXELS
        none = ctypes.c_void_p(0)
        check(lib.lsf_state_prepare(_ptr(live, n, "live"), _ptr(canonical, n, "canonical"),
                                    none if self.sparse_reach else _ptr(self.states[0], 4 * n, "state"),
                                    none if split or self.sparse_reach else _ptr(self.states[1], 4 * n, "state"),
                                    ctypes.byref(grid), ctypes.c_void_p(self._scratch.data_ptr()),
                                    ctypes.c_void_p(totals.data_ptr()), stream_ptr()), "lsf_state_prepare")
        self._totals_host = pinned_scratch("prepare totals", 4, torch.int64)
        self._totals_host.copy_(totals, non_blocking=True)
        self._cuts_host = self._cut_chunks = None
        if cut_chunks is not None and cut_chunks.numel():
            chunks = (n + self.CHUNK - 1) // self.CHUNK
            # the scratch starts with the two arrays of exclusive prefix counts, chunks + 1 apart (lsf_state_prepare)
            prefix = self._scratch[:2 * (chunks + 1)].view(2, chunks + 1)
            cuts = torch.index_select(prefix, 1, cut_chunks)  # indices < chunks + 1: the caller clamps (see collect)
            self._cuts_host = pinned_scratch("prepare cuts", cuts.numel(), torch.int32)
            self._cuts_host.copy_(cuts.view(-1), non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()
        # the states are written BEHIND the copy of the list sizes: the card fills them while the host wakes up on
        # the sizes and enqueues the list fills, instead of idling there.  Sparse: both states, in the chunks near the
        # band only (a quarter of a 256^3 sphere pair: 2 x 64 MB instead of 2 x 256 MB); else the second state (two states
        # in the counting pass: 125 us in front of the sizes; one: 80 us, and these 65 us overlap the host's round trip)
        if self.sparse_reach:
            check(lib.lsf_state_pack_needed(_ptr(live, n, "live"), _ptr(self.states[0], 4 * n, "state"),
                                            _ptr(self.states[1], 4 * n, "state"), ctypes.byref(full_range(grid)),
                                            ctypes.c_void_p(self._scratch.data_ptr()), self.sparse_reach, 0,
                                            stream_ptr()), "lsf_state_pack_needed")
        elif split:
            check(lib.lsf_state_pack(_ptr(live, n, "live"), ctypes.c_void_p(0), _ptr(self.states[1], 4 * n, "state"),
                                     ctypes.c_void_p(0), ctypes.byref(full_range(grid)), stream_ptr()), "lsf_state_pack")

    def complete(self, state, live):
        """sparse states only: write (live, 0) into the chunks of `state` the prepare step left uninitialised (live = the
        input live field, or any array that still holds it at the voxels outside the band lists)"""
        if not self.sparse_reach:
            return state
        n = n_voxels(self.grid)
        check(lib.lsf_state_pack_needed(_ptr(live, n, "live"), _ptr(state, 4 * n, "state"), ctypes.c_void_p(0),
                                        ctypes.byref(full_range(self.grid)), ctypes.c_void_p(self._scratch.data_ptr()),
                                        self.sparse_reach, 1, stream_ptr()), "lsf_state_pack_needed")
        return state

    def needed_fraction(self):
        """share of the 1024-voxel chunks the sparse prepare step initialised (a host read: measurements, tests)"""
        n = n_voxels(self.grid)
        chunks = (n + self.CHUNK - 1) // self.CHUNK
        at = 2 * (chunks + 1) + 64 * chunks + 2 * chunks + chunks  # lsf_slavcheva.hip::prepare_needed
        return float(self._scratch[at:at + chunks].ne(0).float().mean().item())

    def collect(self):
        """(band lists -- empty lists dropped, but never both --, (number of voxels outside the band with
        live = -canonical, the first of them or -1)); the second is what state_finalize_listed needs"""
        self._copied.synchronize()
        counts = self._totals_host.tolist()
        p_scratch = ctypes.c_void_p(self._scratch.data_ptr())
        lists = []
        for k, (subset, count) in enumerate(zip((_lib.BAND_INTERIOR, _lib.BAND_BOUNDARY), counts[:2])):
            if count == 0 and not (k == 1 and not lists):
                continue
            indices = torch.empty(max(count, 1), dtype=torch.int32, device=self._scratch.device)
            if count:
                check(lib.lsf_band_list_fill_prepared(ctypes.byref(self.grid), subset, p_scratch,
                                                      ctypes.c_void_p(indices.data_ptr()), stream_ptr()),
                      "lsf_band_list_fill_prepared")
            lists.append(BandList(indices, count, subset))
        self.cuts = None
        if self._cuts_host is not None:  # {subset: positions}, a chunk index past the last chunk = the list's size
            flat = self._cuts_host.tolist()
            per = len(flat) // 2
            self.cuts = {subset: flat[k * per:(k + 1) * per]
                         for k, subset in enumerate((_lib.BAND_INTERIOR, _lib.BAND_BOUNDARY))}
            self.cut_totals = {_lib.BAND_INTERIOR: counts[0], _lib.BAND_BOUNDARY: counts[1]}
        return lists, (counts[2], counts[3])


def state_prepare(live, canonical, grid=None):
    """StatePrepare in one call: (states, band lists, unlisted counts)"""
    prepared = StatePrepare(live, canonical, grid)
    lists, unlisted = prepared.collect()
    return prepared.states, lists, unlisted


def state_pack(live, warp_planar=None, grid=None, copies=2):
    """(live, planar warp or zeros) -> `copies` identical float4 state tensors [z,]y,x,4 (lsf_state_pack)"""
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    states = [torch.empty(tuple(live.shape) + (4,), dtype=torch.float32, device=live.device) for _ in range(copies)]
    check(lib.lsf_state_pack(_ptr(live, n, "live"), _ptr(warp_planar, n * grid.dims, "warp", allow_none=True),
                             _ptr(states[0], 4 * n, "state"),
                             _ptr(states[1], 4 * n, "state") if copies > 1 else ctypes.c_void_p(0),
                             ctypes.byref(full_range(grid)), stream_ptr()), "lsf_state_pack")
    return states


def state_unpack(state, grid, live_out=None, warp_planar_out=None, warp_interleaved_out=None):
    """float4 state -> live and / or planar warp and / or interleaved warp (lsf_state_unpack), all slices"""
    n = n_voxels(grid)
    check(lib.lsf_state_unpack(_ptr(state, 4 * n, "state"), _ptr(live_out, n, "live_out", allow_none=True),
                               _ptr(warp_planar_out, n * grid.dims, "warp_planar_out", allow_none=True),
                               _ptr(warp_interleaved_out, n * grid.dims, "warp_interleaved_out", allow_none=True),
                               ctypes.byref(full_range(grid)), stream_ptr()), "lsf_state_unpack")


def state_finalize(state, canonical, grid, live_out=None, warp_planar_out=None, warp_interleaved_out=None,
                   lower_threshold=0.0, statistics=False):
    """one pass over the grid's z-range: float4 state -> the caller's fields and, with `statistics`, the raw
    convergence statistics (float64 [16] device tensor: warp statistics [0:8], |canonical - live| statistics [8:16])"""
    n = n_voxels(grid)
    stats = scratch = None
    if statistics:
        stats = torch.empty(16, dtype=torch.float64, device=state.device)
        scratch = torch.empty(int(lib.lsf_state_finalize_scratch_elements(ctypes.byref(grid))), dtype=torch.float64,
                              device=state.device)
    check(lib.lsf_state_finalize(_ptr(state, 4 * n, "state"), _ptr(canonical, n, "canonical", allow_none=not statistics),
                                 _ptr(live_out, n, "live_out", allow_none=True),
                                 _ptr(warp_planar_out, n * grid.dims, "warp_planar_out", allow_none=True),
                                 _ptr(warp_interleaved_out, n * grid.dims, "warp_interleaved_out", allow_none=True),
                                 ctypes.byref(grid), float(lower_threshold),
                                 _ptr(stats, 16, "statistics", dtype=torch.float64, allow_none=True),
                                 _ptr(scratch, scratch.numel() if scratch is not None else 0, "scratch",
                                      dtype=torch.float64, allow_none=True), stream_ptr()), "lsf_state_finalize")
    return stats


def planar_finalize(live, warp_planar, canonical, grid, live_out=None, warp_interleaved_out=None, lower_threshold=0.0,
                    statistics=False):
    """state_finalize for planar final fields (lsf_planar_finalize); live_out None or a tensor other than `live`"""
    n = n_voxels(grid)
    stats = scratch = None
    if statistics:
        stats = torch.empty(16, dtype=torch.float64, device=live.device)
        scratch = torch.empty(int(lib.lsf_state_finalize_scratch_elements(ctypes.byref(grid))), dtype=torch.float64,
                              device=live.device)
    check(lib.lsf_planar_finalize(_ptr(live, n, "live"), _ptr(warp_planar, n * grid.dims, "warp"),
                                  _ptr(canonical, n, "canonical", allow_none=not statistics),
                                  _ptr(live_out, n, "live_out", allow_none=True),
                                  _ptr(warp_interleaved_out, n * grid.dims, "warp_interleaved_out", allow_none=True),
                                  ctypes.byref(grid), float(lower_threshold),
                                  _ptr(stats, 16, "statistics", dtype=torch.float64, allow_none=True),
                                  _ptr(scratch, scratch.numel() if scratch is not None else 0, "scratch",
                                       dtype=torch.float64, allow_none=True), stream_ptr()), "lsf_planar_finalize")
    return stats


def state_finalize_listed(state, canonical, grid, bands, unlisted, live_out=None, warp_interleaved_out=None,
                          lower_threshold=0.0, statistics=False, skip_flag=None, guard=None):
    """state_finalize of whole arrays that visits the voxels of `bands` only (lsf_state_finalize_listed): live_out must
    hold the input live field and warp_interleaved_out zeros already; `unlisted` as state_prepare returned it;
    skip_flag: device address of a word that turns the pass into a no-op when non-zero (tools/chain: the chain kernel's
    violation word);
    guard = (records, count, limit): the pass is a no-op too when one of the first `count` records holds a maximum update
    that is not below `limit` (sparsely initialised states, StatePrepare(sparse_reach=...))"""
    n = n_voxels(grid)
    bands = [b for b in bands if b.count]
    stats = scratch = None
    if statistics:
        stats = torch.empty(16, dtype=torch.float64, device=state.device)
        scratch = torch.empty(int(lib.lsf_state_finalize_scratch_elements(ctypes.byref(grid))) * max(len(bands), 1),
                              dtype=torch.float64, device=state.device)
    pointers = (ctypes.c_void_p * 2)(*[b.pointer for b in bands])
    counts = (ctypes.c_int64 * 2)(*[b.count for b in bands])
    check(lib.lsf_state_finalize_listed(_ptr(state, 4 * n, "state"),
                                        _ptr(canonical, n, "canonical", allow_none=not statistics),
                                        _ptr(live_out, n, "live_out", allow_none=True),
                                        _ptr(warp_interleaved_out, n * grid.dims, "warp_interleaved_out",
                                             allow_none=True),
                                        ctypes.byref(grid), pointers, counts, len(bands), int(unlisted[0]),
                                        int(unlisted[1]), float(lower_threshold),
                                        _ptr(stats, 16, "statistics", dtype=torch.float64, allow_none=True),
                                        _ptr(scratch, scratch.numel() if scratch is not None else 0, "scratch",
                                             dtype=torch.float64, allow_none=True),
                                        ctypes.c_void_p(skip_flag or 0),
                                        ctypes.c_void_p(guard[0].data_ptr() if guard is not None else 0),
                                        int(guard[1]) if guard is not None else 0,
                                        float(guard[2]) if guard is not None else 0.0, stream_ptr()),
          "lsf_state_finalize_listed")
    return stats


BOX_EDGE = 4  # lsf_band_box: 4 x 4 x 4 voxels


def boxes_ok(grid):
    """can lsf_slavcheva_state_iteration_boxes walk this grid?  3-D, extents that are multiples of 4, < 2^28 voxels"""
    return (grid.dims == 3 and grid.nx % BOX_EDGE == 0 and grid.ny % BOX_EDGE == 0 and grid.nz % BOX_EDGE == 0
            and n_voxels(grid) <= 0x0fffffff)


def band_boxes(prepared, subset=_lib.BAND_INTERIOR):
    """the INTERIOR (or, subset = BAND_ALL, all) band voxels of a StatePrepare as boxes (lsf_band_boxes_count / _fill, from
    the ballots the counting pass kept): (int64 tensor [n, 2] = lsf_band_box records, n).  One host read of the count."""
    grid = full_range(prepared.grid)
    scratch = torch.empty(int(lib.lsf_band_boxes_scratch_elements(ctypes.byref(grid))), dtype=torch.int32,
                          device=prepared._scratch.device)
    count = torch.zeros(1, dtype=torch.int64, device=scratch.device)
    p_prepare = ctypes.c_void_p(prepared._scratch.data_ptr())
    check(lib.lsf_band_boxes_count(ctypes.byref(grid), subset, p_prepare, ctypes.c_void_p(scratch.data_ptr()),
                                   ctypes.c_void_p(count.data_ptr()), stream_ptr()), "lsf_band_boxes_count")
    n = int(count.item())
    boxes = torch.empty((max(n, 1), 2), dtype=torch.int64, device=scratch.device)
    if n:
        check(lib.lsf_band_boxes_fill(ctypes.byref(grid), subset, p_prepare, ctypes.c_void_p(scratch.data_ptr()),
                                      ctypes.c_void_p(boxes.data_ptr()), stream_ptr()), "lsf_band_boxes_fill")
    return boxes, n


def band_boxes_canonical(canonical, grid, boxes, n_boxes):
    """the canonical values of the boxes' voxels, 64 per box in box order (lsf_band_boxes_canonical): what the box walk reads"""
    out = torch.empty(max(n_boxes, 1) * BOX_EDGE ** 3, dtype=torch.float32, device=canonical.device)
    check(lib.lsf_band_boxes_canonical(_ptr(canonical, n_voxels(grid), "canonical"), ctypes.byref(grid),
                                       ctypes.c_void_p(boxes.data_ptr()), int(n_boxes), ctypes.c_void_p(out.data_ptr()),
                                       stream_ptr()), "lsf_band_boxes_canonical")
    return out


def slavcheva_state_iteration_boxes(state_in, canonical_boxed, state_out, grid, params, gate, records, index, boxes, n_boxes):
    """the fused iteration over the INTERIOR band voxels, box by box (lsf_slavcheva_state_iteration_boxes); canonical_boxed:
    band_boxes_canonical of the same boxes"""
    n = n_voxels(grid)
    check(lib.lsf_slavcheva_state_iteration_boxes(_ptr(state_in, 4 * n, "state_in"),
                                                  _ptr(canonical_boxed, n_boxes * BOX_EDGE ** 3, "canonical_boxed"),
                                                  _ptr(state_out, 4 * n, "state_out"), ctypes.byref(grid),
                                                  ctypes.byref(params), _gate_ref(gate), _record_ptr(records, index),
                                                  ctypes.c_void_p(boxes.data_ptr()), int(n_boxes), stream_ptr()),
          "lsf_slavcheva_state_iteration_boxes")


def full_range(grid):
    """the same grid with the launch range widened to every allocated slice"""
    g = Grid.from_buffer_copy(grid)
    g.z_begin, g.z_end = 0, g.nz
    return g


def slavcheva_state_iteration(state_in, canonical, state_out, grid, params, gate, records, index, band=None):
    n = n_voxels(grid)
    check(lib.lsf_slavcheva_state_iteration(_ptr(state_in, 4 * n, "state_in"), _ptr(canonical, n, "canonical"),
                                            _ptr(state_out, 4 * n, "state_out"), ctypes.byref(grid),
                                            ctypes.byref(params), _gate_ref(gate), _record_ptr(records, index),
                                            band.pointer if band is not None else ctypes.c_void_p(0),
                                            band.count if band is not None else 0,
                                            band.subset if band is not None else 0, stream_ptr()),
          "lsf_slavcheva_state_iteration")


def slavcheva_update_rewarp(live, canonical, g, warp_out, live_out, grid, params, gate, records, index, band=None):
    n = n_voxels(grid)
    nd = n * grid.dims
    check(lib.lsf_slavcheva_update_rewarp(_ptr(live, n, "live"), _ptr(canonical, n, "canonical"), _ptr(g, nd, "g"),
                                          _ptr(warp_out, nd, "warp_out"), _ptr(live_out, n, "live_out"),
                                          ctypes.byref(grid), ctypes.byref(params), _gate_ref(gate),
                                          _record_ptr(records, index),
                                          band.pointer if band is not None else ctypes.c_void_p(0),
                                          band.count if band is not None else 0, stream_ptr()),
          "lsf_slavcheva_update_rewarp")


def slavcheva_filter_update_rewarp(src, mask_source, live, g_out, warp_out, live_out, grid, params, axis, taps, gate,
                                   records, index, band):
    """last zero-preserving filter pass at the band voxels + update + re-warp in one launch
    (lsf_slavcheva_filter_update_rewarp); taps: 3 / 5 / 7 / 9 (LISTED_TAP_COUNTS)"""
    taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float64))
    n = n_voxels(grid)
    nd = n * grid.dims
    check(lib.lsf_slavcheva_filter_update_rewarp(_ptr(src, nd, "filter src"), _ptr(mask_source, nd, "zero mask"),
                                                 _ptr(live, n, "live"), _ptr(g_out, nd, "g_out"),
                                                 _ptr(warp_out, nd, "warp_out"), _ptr(live_out, n, "live_out"),
                                                 ctypes.byref(grid), ctypes.byref(params), int(axis),
                                                 taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size,
                                                 _gate_ref(gate), _record_ptr(records, index), band.pointer, band.count,
                                                 stream_ptr()), "lsf_slavcheva_filter_update_rewarp")


class IterationLauncher:
    """Pre-validated launch arguments for the per-iteration kernels of one level: tensors are checked ONCE, their
    device pointers, the grid, the parameter block and one gate / record pointer per iteration slot are materialised as
    ctypes objects up front, so that enqueueing an iteration is a bare foreign call (the 2-D levels and coarse 3-D levels
    are launch-bound: ~4 us of kernel against ~15 us of per-call Python otherwise)."""

    def __init__(self, grid, records, gate_mode, gate_a, gate_b=0.0):
        self.grid = grid
        self.grid_ref = ctypes.byref(grid)
        self.records = records
        n = records.shape[0]
        base = records.data_ptr()
        self.record_ptrs = [ctypes.c_void_p(base + i * _lib.RECORD_BYTES) for i in range(n)]
        self.gates = [Gate(base + i * _lib.RECORD_BYTES, int(gate_mode), float(gate_a), float(gate_b))
                      for i in range(n)]
        self.gate_refs = [ctypes.byref(g) for g in self.gates]
        self._keep = []

    def pointer(self, t, numel, name, allow_none=False):
        p = _ptr(t, numel, name, allow_none=allow_none)
        self._keep.append(t)
        return p

    def gate_ref(self, prev_index):
        return None if prev_index is None or prev_index < 0 else self.gate_refs[prev_index]


# ---------------------------------------------------------------------------------------------- a20
def warp_statistics(warp_planar, canonical, live, lower_threshold, grid=None):
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    out = torch.empty(8, dtype=torch.float64, device=live.device)
    check(lib.lsf_warp_statistics(_ptr(warp_planar, n * grid.dims, "warp"), _ptr(canonical, n, "canonical"),
                                  _ptr(live, n, "live"), ctypes.byref(grid), float(lower_threshold),
                                  _ptr(out, 8, "out8", dtype=torch.float64), stream_ptr()), "lsf_warp_statistics")
    return out


def tsdf_difference_statistics(canonical, live, grid=None):
    grid = grid or make_grid(live.shape)
    n = n_voxels(grid)
    out = torch.empty(8, dtype=torch.float64, device=live.device)
    check(lib.lsf_tsdf_difference_statistics(_ptr(canonical, n, "canonical"), _ptr(live, n, "live"),
                                             ctypes.byref(grid), _ptr(out, 8, "out8", dtype=torch.float64),
                                             stream_ptr()), "lsf_tsdf_difference_statistics")
    return out
