"""Torch-facing wrappers around the C ABI of liblsf_hip.so: what every wrapper needs (device.py re-exports it) -- the
current stream and device, pinned scratch, grids, pointer checks, iteration records and gates.

torch is plumbing here: it owns device memory and the HIP stream.  Every wrapper validates dtype, device,
contiguity and element counts on the host BEFORE the launch (a hand-written kernel that reads past a buffer can
take the whole GPU down), then passes raw device pointers + the current HIP stream to the library.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import Gate, Grid, HierParams, SlavchevaParams, check, lib

RECORD_WORDS = _lib.RECORD_BYTES // 8  # an iteration record is 8 slots x 512 int64 words (see lsf_iteration_record)


_gpu_seen = False


def require_gpu():
    global _gpu_seen
    if _gpu_seen:  # (a card does not go away: the check's environment reads are then paid once, not twice per call)
        return
    if not torch.cuda.is_available():
        raise RuntimeError("levelsetfusion-python_amd needs an AMD GPU (ROCm): torch.cuda.is_available() is False. "
                           "There is no CPU execution path in this package.")
    _gpu_seen = True


# torch.cuda.current_stream() / current_device() walk half a dozen Python frames each (device-type look-up, availability
# check, an environment read): ~4 us per call, a dozen calls per optimize() -- cProfile of one call, tools/host_profile.py.
# The two raw queries behind them are single C calls.
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_checked = False


def _check_raw_queries():
    """the two queries are private to torch: the first time a GPU is used they are held against the public calls, and
    dropped for good if they are missing, raise, or answer differently (another torch release)"""
    global _raw_device, _raw_stream, _raw_checked
    _raw_checked = True
    try:
        ok = (_raw_device is not None and _raw_stream is not None
              and _raw_device() == torch.cuda.current_device()
              and _raw_stream(_raw_device()) == torch.cuda.current_stream().cuda_stream)
    except Exception:  # noqa: BLE001 -- any failure means: use the public calls
        ok = False
    if not ok:
        _raw_device = _raw_stream = None


def current_device_index():
    if not _raw_checked:
        _check_raw_queries()
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def current_stream_handle():
    """the current HIP stream of the current device as an integer (hipStream_t)"""
    if not _raw_checked:
        _check_raw_queries()
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def stream_ptr():
    return ctypes.c_void_p(current_stream_handle())


_PINNED = {}


def pinned_scratch(name, numel, dtype):
    """a small page-locked host buffer that lives as long as the process (one per device, STREAM and purpose): the
    landing place of the few numbers a call reads back, without a host allocation per call.  The optimizers are
    single-caller objects (as the reference's are): a buffer is consumed before the next call on the same stream fills it
    again -- and optimizers that run side by side (experiment/multipair.py: pairs in flight, one thread and one stream
    each) never share one."""
    # one buffer per purpose, grown to the largest length asked for
    key = (current_device_index(), current_stream_handle(), name, dtype)
    buf = _PINNED.get(key)
    if buf is None or buf.numel() < int(numel):
        buf = _PINNED[key] = torch.empty(max(int(numel), 2 * buf.numel() if buf is not None else 0), dtype=dtype,
                                         pin_memory=True)
    return buf[:int(numel)]


def make_grid(shape, z_begin=0, z_end=None, z_global_offset=0):
    """shape: spatial extents (ny, nx) or (nz, ny, nx)"""
    shape = tuple(int(s) for s in shape)
    if len(shape) == 2:
        nz, (ny, nx), dims = 1, shape, 2
    elif len(shape) == 3:
        (nz, ny, nx), dims = shape, 3
    else:
        raise ValueError("fields must be 2-D or 3-D, got shape %r" % (shape,))
    z_end = nz if z_end is None else z_end
    if not (0 <= z_begin <= z_end <= nz):
        raise ValueError("bad z range [%d, %d) for nz = %d" % (z_begin, z_end, nz))
    return Grid(dims, nz, ny, nx, z_begin, z_end, z_global_offset, 0)


def n_voxels(grid):
    return grid.nz * grid.ny * grid.nx


def _ptr(t, numel, name, dtype=torch.float32, allow_none=False):
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0)
        raise ValueError("%s: tensor required" % name)
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s: expected a CUDA/ROCm tensor" % name)
    if t.device.index != current_device_index():
        # kernels are launched on the CURRENT device's current stream (stream_ptr): a tensor that lives elsewhere
        # would be reached through a peer mapping at best
        raise ValueError("%s is on %s but the current device is cuda:%d (torch.cuda.set_device / torch.cuda.device)"
                         % (name, t.device, torch.cuda.current_device()))
    if t.dtype != dtype:
        raise ValueError("%s: expected dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s: tensor must be contiguous" % name)
    if t.numel() != numel:
        raise ValueError("%s: expected %d elements, got %d" % (name, numel, t.numel()))
    return ctypes.c_void_p(t.data_ptr())


def _record_ptr(records, index, name="records"):
    """records: int64 tensor [n, 4] (raw 32-byte records); returns pointer to record `index`"""
    if records.dtype != torch.int64 or not records.is_cuda or not records.is_contiguous() or records.dim() != 2 \
            or records.shape[1] != RECORD_WORDS:
        raise ValueError("%s: expected a contiguous CUDA int64 tensor of shape [n, %d]" % (name, RECORD_WORDS))
    if not (0 <= index < records.shape[0]):
        raise IndexError("%s: record index %d out of range [0, %d)" % (name, index, records.shape[0]))
    return ctypes.c_void_p(records.data_ptr() + index * _lib.RECORD_BYTES)


def new_records(n, device):
    return torch.zeros((n, RECORD_WORDS), dtype=torch.int64, device=device)


def make_gate(records, prev_index, mode, a, b=0.0):
    """gate on record `prev_index` (None / negative: always run)"""
    if records is None or prev_index is None or prev_index < 0:
        return None
    return Gate(_record_ptr(records, prev_index).value, int(mode), float(a), float(b))


def _gate_ref(gate):
    return ctypes.byref(gate) if gate is not None else None


def slot_view(records):
    """[n, RECORD_WORDS] (tensor or array) -> [n, slots, SLOT_WORDS]: word 0 = packed max, words 1..3 = energies"""
    return records.reshape(records.shape[0], _lib.RECORD_SLOTS, _lib.SLOT_WORDS)


def set_record_max(records, index, packed):
    """make record `index` read as `packed` to a gate (slot 0 holds it, the other slots are cleared)"""
    records[index].zero_()
    records[index, 0] = packed


USED_SLOT_WORDS = 4  # packed max + three energies; the rest of a slot is padding (slots sit 4 KiB apart)


def records_to_host(records):
    """device records [n, RECORD_WORDS] -> numpy int64 [n, slots, 4]: only the used words cross PCIe, into a page-locked
    buffer (no staging copy); the call returns when they have arrived"""
    used = slot_view(records)[:, :, :USED_SLOT_WORDS].contiguous()
    if not used.is_cuda:
        return used.numpy()
    host = pinned_scratch("records", used.numel(), used.dtype)
    host.copy_(used.view(-1), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy().reshape(tuple(used.shape)).copy()


def decode_records(records_host):
    """records_host: numpy int64 [n, slots, >= 4] (records_to_host) or [n, RECORD_WORDS] -> dict of arrays (max value,
    linear arg-max index, energies, executed); combines the slots of every record: max of the packed maxima, sum of the
    energies"""
    raw = records_host
    if raw.ndim == 2:
        raw = slot_view(np.ascontiguousarray(raw))
    if raw.dtype == np.int64 and raw.ndim == 3 and raw.shape[2] >= USED_SLOT_WORDS and raw.flags.c_contiguous:
        # one host call into the library (lsf_records_decode) instead of a dozen numpy calls: ~30 -> ~8 us per optimize()
        n = raw.shape[0]
        max_value, index = np.empty(n, np.float32), np.empty(n, np.int64)
        energies, executed = np.empty((n, 3), np.float64), np.empty(n, np.bool_)
        check(lib.lsf_records_decode(raw.ctypes.data, n, raw.shape[1], raw.shape[2], max_value.ctypes.data,
                                     index.ctypes.data, energies.ctypes.data, executed.ctypes.data), "lsf_records_decode")
        return dict(executed=executed, max_value=max_value, argmax=index, data_energy=energies[:, 0],
                    smoothing_energy=energies[:, 1], level_set_energy=energies[:, 2])
    words = raw.view(np.uint64)  # same item size: no copy, strides kept
    packed = words[:, :, 0].max(axis=1)
    energies = np.ascontiguousarray(raw[:, :, 1:4]).view(np.float64).sum(axis=1)
    executed = packed != 0
    max_value = (packed >> np.uint64(32)).astype(np.uint32).view(np.float32)
    index = (~packed.astype(np.uint32)).astype(np.int64)
    return dict(executed=executed, max_value=max_value, argmax=index, data_energy=energies[:, 0],
                smoothing_energy=energies[:, 1], level_set_energy=energies[:, 2])
