"""The CHAIN kernel -- K fused iterations of an INTERIOR band list in ONE launch, workgroups that wait for their neighbouring
list chunks only (lsf_slavcheva_chain.hip, C ABI lsf_hip_chain.h) -- as a measurement tool.  Round 3 built it into the
product as an opt-in add-on library; it measured 4 % slower than one launch per iteration at 256^3 and 512^3 (DESIGN.md
section 7, round 3), so since round 5 it lives here, outside the package: this module compiles it (hipcc, gfx950) into
tools/chain/bin/liblsf_chain.so against the package's own kernel headers, binds it with ctypes and drives it
(StateChain); chain_time.py holds it against per-iteration launches bit for bit and times both, chain_trace.py reads its
in-kernel clocks.  Reference loop: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330, :360-362."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import levelsetfusion_python_amd  # noqa: E402,F401
from levelsetfusion_python_amd import _build, _lib, device as dev  # noqa: E402
from levelsetfusion_python_amd._lib import Grid, SlavchevaParams  # noqa: E402

LIB_PATH = os.path.join(HERE, "bin", "liblsf_chain.so")
ERR_NOT_RESIDENT = -6
_P, _vp, _i32, _i64 = ctypes.POINTER, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
PROTOTYPES = {
    "lsf_state_chain_scratch_elements": (ctypes.c_int64, [_i64, _i32]),
    "lsf_state_chain_shape": (ctypes.c_int, [_i64, _i32, _P(_i32)]),
    "lsf_state_chain_plan": (ctypes.c_int, [_P(Grid), _vp, _i64, _i32, _vp, _vp]),
    "lsf_slavcheva_state_chain": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _vp, _vp, _i64, _i32,
                                                 _i32, _vp, _vp]),
}


def build(extra=(), out=LIB_PATH):
    """hipcc --offload-arch=gfx950 lsf_slavcheva_chain.hip -> liblsf_chain.so (the flags of the product library)"""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(HERE, "lsf_slavcheva_chain.hip")
    deps = [src, os.path.join(HERE, "lsf_hip_chain.h")] + [os.path.join(_build.CSRC, h) for h in _build.HEADERS]
    if os.path.exists(out) and not extra and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [_build.find_hipcc()] + _build.HIPCC_FLAGS + ['-DLSF_BUILD_ID="chain"', '-DLSF_ABI_HASH="%s"' % _build.abi_hash(),
                                                      "-I", _build.CSRC, "-I", os.path.join(ROOT, "include"), "-I", HERE,
                                                      "-shared"] + list(extra) + [src, "-o", out, "-ldl"]
    subprocess.check_call(cmd)
    return out


_handle = None


def chain_lib():
    global _handle
    if _handle is None:
        handle = ctypes.CDLL(os.environ.get("LSF_CHAIN_LIBRARY") or build())
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = restype, argtypes
        _handle = handle
    return _handle


class StateChain:
    """K fused iterations of an INTERIOR band list per launch (lsf_slavcheva_state_chain): the dependency windows of the
    list are planned once (lsf_state_chain_plan), every launch(first, count) then runs iterations first .. first + count - 1
    of the call -- iteration j reads states[j % 2], writes the other, reduces into records[j].
    REACH_LIMIT: update length from which the windows no longer cover the re-warp gather (the launch then raises its
    violation word, which a finalize pass given violation_ptr honours, and the records' maxima tell the host)."""

    REACH_LIMIT = 2.0

    def __init__(self, states, canonical, grid, params, records, band, stages=1):
        if band.subset != _lib.BAND_INTERIOR or not band.count:
            raise ValueError("the chain kernel walks a non-empty INTERIOR band list")
        self.grid = dev.full_range(grid)
        n = dev.n_voxels(grid)
        self.stages = int(stages)
        self._keep = (states, canonical, records, band, params)
        self._states = [dev._ptr(t, 4 * n, "state") for t in states]
        self._canonical = dev._ptr(canonical, n, "canonical")
        self._params = ctypes.byref(params)
        self._records = records
        self._band = band
        self._lib = clib = chain_lib()  # the optional add-on library (include/lsf_hip_chain.h)
        words = int(clib.lsf_state_chain_scratch_elements(band.count, self.stages))
        self.scratch = torch.empty(words, dtype=torch.int32, device=states[0].device)
        self._scratch = ctypes.c_void_p(self.scratch.data_ptr())
        self.violation_ptr = self.scratch.data_ptr() + 4
        shape = (ctypes.c_int32 * 4)()
        _lib.check(clib.lsf_state_chain_shape(band.count, self.stages, shape), "lsf_state_chain_shape")
        self.workgroups, self.stages_used, self.chunks, self.units = (int(v) for v in shape)
        _lib.check(clib.lsf_state_chain_plan(ctypes.byref(self.grid), band.pointer, band.count, self.stages, self._scratch,
                                       dev.stream_ptr()), "lsf_state_chain_plan")

    def launch(self, first, count):
        """False: the kernel's workgroups cannot all be resident on this device (nothing was launched)"""
        a, b = self._states[first % 2], self._states[(first + 1) % 2]
        status = self._lib.lsf_slavcheva_state_chain(a, b, self._canonical, ctypes.byref(self.grid), self._params,
                                               dev._record_ptr(self._records, first), self._band.pointer, self._band.count,
                                               int(count), self.stages, self._scratch, dev.stream_ptr())
        if status == ERR_NOT_RESIDENT:
            return False
        _lib.check(status, "lsf_slavcheva_state_chain")
        return True

    def aborted(self):
        """True when a wait of the last launch timed out (control word 0; a host read: call it behind the records)"""
        return bool(int(self.scratch[0].item()))


if __name__ == "__main__":
    print(build())
