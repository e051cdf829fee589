"""What a SlavchevaEngine.optimize() call leaves behind: the final fields on the device (handed out on demand) and the
per-iteration log."""
import torch

from . import device as dev


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
