"""The SobolevFusion iteration (gradient, zero-preserving separable filter, update + re-warp:
slavcheva_optimizer2d.py:163-236 with math_utils/convolution.py:114-132): the launch plan on the float4 layouts (band
lists / boxes of a whole volume) and the planar-field iteration z-slabs and list-less runs keep."""
import ctypes

import numpy as np
import torch

from . import _lib, device as dev
from .engine_common import _conv_axis_order


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



class PlanarSobolevMixin:
    """SlavchevaEngine's iteration on PLANAR fields (z-slabs, filters of other lengths, list-less runs)"""

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
