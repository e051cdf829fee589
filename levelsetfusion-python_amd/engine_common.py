"""What the engine modules share: input conversion, the pyramid's level-count rule, the filter's axis order, the
HIP-graph capture lock, small helpers."""
import threading

import numpy as np
import torch

from . import device as dev


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

