"""the reference's "focus neighbourhood" trace as an opt-in probe of the Slavcheva loop

slavcheva_optimizer2d.py:48-55 (VoxelLog), :319-322 (what is appended every iteration), :422-430 (the 3 x 3 voxels around
the focus coordinate).  The reference fills it on every call (its focus coordinate is a module global,
utils/sampling.py:27-33); here nothing is traced unless `focus_voxels` is set ("opt-in hooks that force a
device->host copy only when enabled"): a traced call synchronises every iteration and re-derives what the fused kernels
never store -- the update BEFORE the snap of warp_field_advanced zeroes it."""
import numpy as np
import torch

from . import _lib
from . import device as dev
from .engine_common import _conv_axis_order


class FocusMixin:
    focus_voxels = None  # index tuples in the arrays' axis order ((y, x) / (z, y, x)), or None: nothing is traced
    focus_trace = None   # what the last traced call saw: dict(canonical=[V], sdf=[iterations][V], warp=[iterations][V])

    def _focus_begin(self, live, canonical):
        """the traced voxels of this call, as flat indices on the device, and their canonical values (:353-354)"""
        if self._slab():
            raise ValueError("the focus-neighbourhood trace reads single voxels of the whole volume: not on a z-slab rank")
        flat = np.ravel_multi_index(np.asarray(self.focus_voxels, dtype=np.int64).T, tuple(live.shape))
        self._focus_flat = torch.as_tensor(np.atleast_1d(flat), dtype=torch.int64, device=live.device)
        self.focus_trace = dict(canonical=canonical.reshape(-1)[self._focus_flat].cpu().numpy(), sdf=[], warp=[])

    def _probe_focus(self, i, lives, warps, states, canonical, grid):
        """iteration i has run: the live value it started from and the length of its update BEFORE the snap, at the traced
        voxels.  The inputs of the iteration are still in the ping-pong buffers; the (filtered) gradient is recomputed
        from them by the unfused kernels on the whole array -- the same code path as gradient_field(), without the update
        kernel that zeroes it where the live field snapped."""
        if states is not None:
            live_in = torch.empty(tuple(states[0].shape[:-1]), dtype=torch.float32, device=states[0].device)
            warp_in = torch.empty((grid.dims,) + tuple(live_in.shape), dtype=torch.float32, device=live_in.device)
            dev.state_unpack(states[i % 2], dev.full_range(grid), live_in, warp_in, None)
        else:
            live_in, warp_in = lives[i % 2], warps[i % 2]
        params = _lib.SlavchevaParams.from_buffer_copy(self.params)
        params.energy_mode = _lib.ENERGY_NONE
        g0 = torch.empty_like(warp_in)
        dev.slavcheva_gradient(live_in, canonical, warp_in, g0, grid, params, None, dev.new_records(1, live_in.device), 0)
        g = g0
        if self.sobolev:
            spare = [torch.empty_like(g0), torch.empty_like(g0)]
            for k, axis in enumerate(_conv_axis_order(grid.dims)):
                dev.convolve_axis(g, spare[k % 2], g0, grid, axis, self.sobolev_kernel)
                g = spare[k % 2]
        at = g.reshape(grid.dims, -1)[:, self._focus_flat].cpu().numpy()  # [D, V]; component 0 = x
        update = ((-at).astype(np.float32) * np.float32(self.params.rate)).astype(np.float32)
        squares = (update[0] * update[0]).astype(np.float32)
        for c in range(1, grid.dims):
            squares = (squares + update[c] * update[c]).astype(np.float32)
        self.focus_trace["warp"].append(np.sqrt(squares).astype(np.float32))
        self.focus_trace["sdf"].append(live_in.reshape(-1)[self._focus_flat].cpu().numpy())
