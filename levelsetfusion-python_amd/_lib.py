"""ctypes binding of liblsf_hip.so (C ABI: include/lsf_hip.h).

There is NO fallback: if the HIP library is missing, importing this module raises.  The numpy oracle under
oracle/ is test infrastructure and is never imported from the package.
"""
import ctypes
import os

# torch FIRST: the PyTorch-ROCm wheel ships its own libamdhip64.so; liblsf_hip.so must bind to that already-loaded
# HIP runtime (same soname), otherwise the process ends up with two runtimes and launches on torch's streams fail
# with hipErrorNoDevice.
import torch  # noqa: F401

from ._build import LIB_PATH

c_float_p = ctypes.c_void_p  # device pointers travel as integers (tensor.data_ptr())

MAX_KERNEL_TAPS = 31
ABI_VERSION = 4
# _build.abi_hash() of the include/lsf_hip.h THIS binding was written against (structures and prototypes below mirror
# it).  The library carries the hash of the header it was compiled from (lsf_abi_hash()); a mismatch is refused at load
# time, and tests/test_cabi_and_host.py checks this constant against the header in the tree -- so editing a struct or
# a prototype in the header without revisiting the binding fails on the CPU, and a stale or variant .so cannot be
# called through structures of another shape.
HEADER_ABI_HASH = "2b52a5b7ac923a65"

ERRORS = {-1: "LSF_ERR_BAD_ARGUMENT", -2: "LSF_ERR_BAD_DIMS", -3: "LSF_ERR_KERNEL_TOO_LONG",
          -4: "LSF_ERR_RCCL_UNAVAILABLE", -5: "LSF_ERR_RCCL_FAILED", -6: "LSF_ERR_NOT_RESIDENT"}

SMOOTHING_TIKHONOV, SMOOTHING_KILLING = 0, 1
DATA_BASIC, DATA_THRESHOLDED_FDM = 0, 2
ENERGY_NONE, ENERGY_DIRECT, ENERGY_VECTORIZED = 0, 1, 2
GATE_HIERARCHICAL, GATE_SLAVCHEVA, GATE_OPEN = 0, 1, 2


class Grid(ctypes.Structure):
    _fields_ = [("dims", ctypes.c_int32), ("nz", ctypes.c_int32), ("ny", ctypes.c_int32), ("nx", ctypes.c_int32),
                ("z_begin", ctypes.c_int32), ("z_end", ctypes.c_int32), ("z_global_offset", ctypes.c_int32),
                ("y_global_offset", ctypes.c_int32), ("energy_z_begin", ctypes.c_int32), ("energy_z_end", ctypes.c_int32),
                ("ny_global", ctypes.c_int32), ("energy_y_begin", ctypes.c_int32), ("energy_y_end", ctypes.c_int32)]


BAND_ALL, BAND_INTERIOR, BAND_BOUNDARY = 0, 1, 2
RECORD_SLOTS = 8


class RecordSlot(ctypes.Structure):
    _fields_ = [("max_packed", ctypes.c_uint64), ("data_energy", ctypes.c_double),
                ("smoothing_energy", ctypes.c_double), ("level_set_energy", ctypes.c_double),
                ("pad", ctypes.c_uint64 * 508)]


class IterationRecord(ctypes.Structure):
    """8 partial records 4 KiB apart (one per XCD: separate memory channels); value = max over the slots' max_packed,
    sum over their energies"""
    _fields_ = [("slot", RecordSlot * RECORD_SLOTS)]


RECORD_BYTES = ctypes.sizeof(IterationRecord)  # 32768
SLOT_WORDS = ctypes.sizeof(RecordSlot) // 8    # 512


class Gate(ctypes.Structure):
    _fields_ = [("prev_record", ctypes.c_void_p), ("mode", ctypes.c_int32), ("a", ctypes.c_float),
                ("b", ctypes.c_float)]


SLAB_LAUNCH, SLAB_EXCHANGE, SLAB_EXCHANGE_DEFERRED, SLAB_RESUME = 0, 1, 2, 3


class SlabLayoutC(ctypes.Structure):
    _fields_ = [("nz", ctypes.c_int32), ("ny", ctypes.c_int32), ("nx", ctypes.c_int32), ("z_begin", ctypes.c_int32),
                ("z_end", ctypes.c_int32), ("halo", ctypes.c_int32), ("lo_rank", ctypes.c_int32),
                ("hi_rank", ctypes.c_int32)]


class SlabPart(ctypes.Structure):
    _fields_ = [("grid", Grid), ("band_list", ctypes.c_void_p * 2), ("band_count", ctypes.c_int64 * 2),
                ("band_subset", ctypes.c_int32 * 2), ("n_lists", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class SlabFaces(ctypes.Structure):
    _fields_ = [("send_list", ctypes.c_void_p * 2), ("recv_list", ctypes.c_void_p * 2),
                ("send_msg", ctypes.c_void_p * 2), ("recv_msg", ctypes.c_void_p * 2),
                ("send_count", ctypes.c_int64 * 2), ("recv_count", ctypes.c_int64 * 2)]


class BandBox(ctypes.Structure):
    """lsf_band_box: a 4 x 4 x 4 box of the volume and which of its voxels are INTERIOR band voxels"""
    _fields_ = [("origin", ctypes.c_int32), ("reserved", ctypes.c_int32), ("mask", ctypes.c_uint64)]


class StateRun(ctypes.Structure):
    """lsf_state_run: the buffers of a whole fixed-count call enqueued by the library (lsf_state_run_begin / _finish)"""
    _fields_ = [("live", ctypes.c_void_p), ("canonical", ctypes.c_void_p), ("state", ctypes.c_void_p * 2),
                ("prepare_scratch", ctypes.c_void_p), ("totals_device", ctypes.c_void_p),
                ("totals_host", ctypes.c_void_p), ("grid", Grid), ("sparse_reach", ctypes.c_int32),
                ("second_state_late", ctypes.c_int32), ("box_scratch", ctypes.c_void_p), ("box_all", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class StateRunResult(ctypes.Structure):
    _fields_ = [("max_value", ctypes.c_void_p), ("argmax", ctypes.c_void_p), ("energies3", ctypes.c_void_p),
                ("executed", ctypes.c_void_p), ("final_state", ctypes.c_int32), ("n_lists", ctypes.c_int32),
                ("reach_exceeded", ctypes.c_int32), ("compact_faces", ctypes.c_int32)]


class RunLoop(ctypes.Structure):
    """lsf_run_loop: the loop condition of slavcheva_optimizer2d.py:360-362 for lsf_state_run_finish"""
    _fields_ = [("min_iterations", ctypes.c_int32), ("max_iterations", ctypes.c_int32), ("lower_threshold", ctypes.c_float),
                ("upper_threshold", ctypes.c_float), ("check_interval", ctypes.c_int32), ("reserved", ctypes.c_int32)]


SLAB_MAX_CUTS = 72


class SlabRun(ctypes.Structure):
    """lsf_slab_run: a whole fixed-count call of a z-slab rank enqueued by the library (lsf_slab_run_begin / _finish)"""
    _fields_ = [("base", StateRun), ("layout", SlabLayoutC), ("exchange_interval", ctypes.c_int32),
                ("n_cuts", ctypes.c_int32), ("cut_slices", ctypes.c_int32 * SLAB_MAX_CUTS),
                ("cut_entries", (ctypes.c_int64 * SLAB_MAX_CUTS) * 2), ("out_index_entries", ctypes.c_int64),
                ("out_face_entries", ctypes.c_int64)]


class HierParams(ctypes.Structure):
    _fields_ = [("data_term_amplifier", ctypes.c_float), ("tikhonov_strength", ctypes.c_float),
                ("rate", ctypes.c_float), ("tikhonov_enabled", ctypes.c_int32), ("apply_update", ctypes.c_int32),
                ("compute_energy", ctypes.c_int32), ("previous_max", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("packed_nz", ctypes.c_int32), ("packed_z_global_offset", ctypes.c_int32)]


class SlavchevaParams(ctypes.Structure):
    _fields_ = [("isomorphic_enforcement_factor_f64", ctypes.c_double), ("rate", ctypes.c_float),
                ("data_term_weight", ctypes.c_float), ("smoothing_term_weight", ctypes.c_float),
                ("level_set_term_weight", ctypes.c_float), ("isomorphic_enforcement_factor", ctypes.c_float),
                ("killing_c1", ctypes.c_float), ("smoothing_method", ctypes.c_int32),
                ("data_method", ctypes.c_int32), ("level_set_enabled", ctypes.c_int32),
                ("energy_mode", ctypes.c_int32), ("zero_gradient_on_snap", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class TsdfParams(ctypes.Structure):
    _fields_ = [("intrinsics", ctypes.c_double * 4), ("depth_unit_ratio", ctypes.c_double),
                ("voxel_size", ctypes.c_double), ("narrow_band_half_width", ctypes.c_double),
                ("extrinsic", ctypes.c_float * 16), ("array_offset", ctypes.c_int32 * 3),
                ("image_width", ctypes.c_int32), ("image_height", ctypes.c_int32),
                ("image_y_coordinate", ctypes.c_int32), ("default_value", ctypes.c_float),
                ("intrinsics_are_f32", ctypes.c_int32)]


class EwaParams(ctypes.Structure):
    _fields_ = [("covariance_camera_space", ctypes.c_double * 9), ("squared_radius_threshold", ctypes.c_double),
                ("intrinsic_matrix", ctypes.c_float * 9), ("method", ctypes.c_int32)]


_P = ctypes.POINTER
_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); every symbol include/lsf_hip.h declares
PROTOTYPES = {
    "lsf_abi_version": (ctypes.c_int, []),
    "lsf_target_arch": (ctypes.c_char_p, []),
    "lsf_build_id": (ctypes.c_char_p, []),
    "lsf_abi_hash": (ctypes.c_char_p, []),
    "lsf_deinterleave": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "lsf_interleave": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "lsf_halo_copy": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _i32, _i32, _i32, _i32, _i32, _vp]),
    "lsf_warp_field": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _f32, _vp]),
    "lsf_warp_field_advanced": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _P(Grid), _i32, _vp]),
    "lsf_pack_live_gradient": (ctypes.c_int, [_vp, _vp, _P(Grid), _vp]),
    "lsf_restrict_mean": (ctypes.c_int, [_vp, _vp, _P(Grid), _i32, _vp]),
    "lsf_prolong_repeat": (ctypes.c_int, [_vp, _vp, _P(Grid), _vp]),
    "lsf_upsample2x_linear": (ctypes.c_int, [_vp, _vp, _P(Grid), _i32, _vp]),
    "lsf_downsample2x_linear": (ctypes.c_int, [_vp, _vp, _P(Grid), _i32, _vp]),
    "lsf_convolve_axis": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _i32, _i32, _P(ctypes.c_double), _i32,
                                         _P(Gate), _vp]),
    "lsf_convolve_xy": (ctypes.c_int, [_vp, _vp, _P(Grid), _i32, _P(ctypes.c_double), _i32, _P(Gate), _vp]),
    "lsf_convolve_axis_update": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _P(Grid), _i32, _i32, _P(ctypes.c_double),
                                                _i32, _P(Gate), _vp]),
    "lsf_convolve_xyz": (ctypes.c_int, [_vp, _vp, _vp, _f32, _P(Grid), _i32, _P(ctypes.c_double), _i32, _P(Gate),
                                        _vp]),
    "lsf_hier_iteration": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _P(Grid), _P(HierParams), _P(Gate), _vp, _vp]),
    "lsf_hier_update": (ctypes.c_int, [_vp, _vp, _P(Grid), _f32, _P(Gate), _vp, _vp]),
    "lsf_hier_level_run_2d": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _P(Grid), _P(HierParams), _P(ctypes.c_double),
                                            _i32, _vp, _i32, _i32, ctypes.c_float, _vp]),
    "lsf_slavcheva_gradient": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _P(Gate), _vp,
                                              _vp, _i64, _vp]),
    "lsf_convolve_axis_listed": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _i32, _i32, _P(ctypes.c_double), _i32,
                                                _P(Gate), _vp, _i64, _vp]),
    "lsf_state_prepare_scratch_elements": (ctypes.c_int64, [_P(Grid)]),
    "lsf_band_list_fill_prepared": (ctypes.c_int, [_P(Grid), _i32, _vp, _vp, _vp]),
    "lsf_state_prepare": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _vp, _vp, _vp]),
    "lsf_state_pack": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _vp]),
    "lsf_state_pack_needed": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _vp, _i32, _i32, _vp]),
    "lsf_state_unpack": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _vp]),
    "lsf_state_finalize_scratch_elements": (ctypes.c_int64, [_P(Grid)]),
    "lsf_state_finalize": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _P(Grid), _f32, _vp, _vp, _vp]),
    "lsf_planar_finalize": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _P(Grid), _f32, _vp, _vp, _vp]),
    "lsf_state_finalize_listed": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _P(ctypes.c_void_p), _P(_i64), _i32,
                                                 _i64, _i64, _f32, _vp, _vp, _vp, _vp, _i32, _f32, _vp]),
    "lsf_slavcheva_state_iteration": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _P(Gate), _vp,
                                                     _vp, _i64, _i32, _vp]),
    "lsf_band_boxes_scratch_elements": (ctypes.c_int64, [_P(Grid)]),
    "lsf_band_boxes_count": (ctypes.c_int, [_P(Grid), ctypes.c_int32, _vp, _vp, _vp, _vp]),
    "lsf_band_boxes_fill": (ctypes.c_int, [_P(Grid), ctypes.c_int32, _vp, _vp, _vp, _vp]),
    "lsf_band_boxes_canonical": (ctypes.c_int, [_vp, _P(Grid), _vp, ctypes.c_int64, _vp, _vp]),
    "lsf_slavcheva_state_iteration_boxes": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _P(Gate), _vp, _vp,
                                                           _i64, _vp]),
    "lsf_state_run_begin": (ctypes.c_int, [_P(StateRun), _vp]),
    "lsf_state_run_finish": (ctypes.c_int, [_P(StateRun), _P(SlavchevaParams), _vp, _vp, _vp, _vp, _vp, _i32, _P(RunLoop), _vp,
                                            _f32, _vp, _vp, _vp, _vp, _P(StateRunResult), _vp]),
    "lsf_sobolev_run_finish": (ctypes.c_int, [_P(StateRun), _P(SlavchevaParams), _P(ctypes.c_double), _i32, _vp, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _i32, _P(RunLoop), _vp, _f32, _vp, _vp, _vp, _vp,
                                              _P(StateRunResult), _vp]),
    "lsf_slab_run_begin": (ctypes.c_int, [_P(SlabRun), _vp]),
    "lsf_slab_run_finish": (ctypes.c_int, [_P(SlabRun), _vp, _P(SlavchevaParams), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                           _P(StateRunResult), _vp]),
    "lsf_band_scratch_elements": (ctypes.c_int64, [_P(Grid)]),
    "lsf_band_count": (ctypes.c_int, [_vp, _vp, _P(Grid), _i32, _vp, _vp, _vp]),
    "lsf_band_list_fill": (ctypes.c_int, [_vp, _vp, _P(Grid), _i32, _vp, _vp, _vp]),
    "lsf_records_decode": (ctypes.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "lsf_merge_sorted_runs": (ctypes.c_int, [_P(_vp), _P(_i64), _P(_vp), _P(_i64), _P(_vp), _i32, _vp]),
    "lsf_slab_unique_id": (ctypes.c_int, [ctypes.c_char_p, _vp]),
    "lsf_slab_comm_create": (ctypes.c_int, [ctypes.c_char_p, _vp, _i32, _i32, _P(_vp)]),
    "lsf_slab_comm_destroy": (ctypes.c_int, [_vp]),
    "lsf_slab_comm_info": (ctypes.c_int, [_vp, _P(_i32), _P(_i32)]),
    "lsf_slab_face_counts_begin": (ctypes.c_int, [_vp, _P(_i64)]),
    "lsf_slab_face_counts_end": (ctypes.c_int, [_vp, _P(_i64)]),
    "lsf_slab_state_iteration": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(SlabLayoutC), _P(SlabPart), _i32, _P(SlabPart),
                                                _i32, _P(SlavchevaParams), _P(Gate), _vp, _i32, _P(SlabFaces), _vp]),
    "lsf_slavcheva_filter_update_rewarp": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _P(Grid), _P(SlavchevaParams),
                                                          _i32, _P(ctypes.c_double), _i32, _P(Gate), _vp, _vp, _i64,
                                                          _vp]),
    "lsf_sobolev_state_gradient": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _P(Gate), _vp, _vp, _i64,
                                                  _vp]),
    "lsf_band_list_strip_major": (ctypes.c_int, [_vp, _i64, _P(Grid), _i32, _vp, _vp, _vp]),
    "lsf_sobolev_state_gradient_x": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _P(ctypes.c_double), _i32,
                                                    _P(Gate), _vp, _vp, _i64, _i32, _vp]),
    "lsf_zero_listed4": (ctypes.c_int, [_vp, _P(Grid), _vp, _i64, _i32, _vp]),
    "lsf_convolve_axis_listed4": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _i32, _P(ctypes.c_double), _i32, _P(Gate), _vp,
                                                 _i64, _vp]),
    "lsf_sobolev_state_update": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _P(Grid), _P(SlavchevaParams), _i32,
                                                _P(ctypes.c_double), _i32, _P(Gate), _vp, _vp, _i64, _i32, _vp]),
    "lsf_sobolev_state_update_boxes": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Grid), _P(SlavchevaParams),
                                                      _P(ctypes.c_double), _i32, _P(Gate), _vp, _vp, _i64, _vp]),
    "lsf_slavcheva_update_rewarp": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _P(Grid), _P(SlavchevaParams),
                                                   _P(Gate), _vp, _vp, _i64, _vp]),
    "lsf_warp_statistics": (ctypes.c_int, [_vp, _vp, _vp, _P(Grid), _f32, _vp, _vp]),
    "lsf_tsdf_difference_statistics": (ctypes.c_int, [_vp, _vp, _P(Grid), _vp, _vp]),
    "lsf_tsdf_generate_nearest": (ctypes.c_int, [_vp, _vp, _P(Grid), _P(TsdfParams), _vp]),
    "lsf_tsdf_generate_bilinear": (ctypes.c_int, [_vp, _vp, _P(Grid), _P(TsdfParams), _i32, _vp]),
    "lsf_tsdf_generate_ewa": (ctypes.c_int, [_vp, _vp, _P(Grid), _P(TsdfParams), _P(EwaParams), _vp]),
}


class LsfHipError(RuntimeError):
    pass


def _load():
    # LSF_HIP_LIBRARY: another build of the SAME HIP library (A/B measurements of kernel variants, tools/ab_state_kernel.py)
    path = os.environ.get("LSF_HIP_LIBRARY") or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            "liblsf_hip.so is missing (%s).  This package has no CPU fallback: build the HIP library first with\n"
            "    python -c 'import __graft_entry__ as g; g.build()'      (needs hipcc, cross-compiles gfx950)"
            % path)
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        if argtypes is None:
            continue
        fn = getattr(lib, name)  # AttributeError here = the .so is stale against the header
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lsf_abi_version() != ABI_VERSION:
        raise ImportError("liblsf_hip.so ABI version %d != binding version %d: rebuild" %
                          (lib.lsf_abi_version(), ABI_VERSION))
    built_against = lib.lsf_abi_hash().decode()
    if built_against != HEADER_ABI_HASH:
        raise ImportError("%s was compiled from an include/lsf_hip.h whose structures / prototypes hash to %s; this binding "
                          "was written against %s: rebuild the library (or update the binding to the header)"
                          % (path, built_against, HEADER_ABI_HASH))
    return lib


lib = _load()

def check(status, what):
    if status == 0:
        return
    if status in ERRORS:
        raise LsfHipError("%s: %s" % (what, ERRORS[status]))
    raise LsfHipError("%s: HIP error %d" % (what, status))
