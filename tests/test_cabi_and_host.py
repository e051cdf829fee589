"""CPU-side checks (no GPU needed): the C-ABI library builds for gfx950, loads, and exports every symbol that
include/lsf_hip.h declares; ctypes structures match the header's layouts; host-side logic (pyramid rules, slab
layout, record decoding, Sobolev filter generation); and the product refuses to run without a GPU instead of
falling back to a CPU path."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lsf_hip.h")


@pytest.fixture(scope="module")
def pkg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_lsf_build_t", os.path.join(ROOT, "levelsetfusion-python_amd",
                                                                              "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(force=False, verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    import levelsetfusion_python_amd as lsf
    return lsf


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    names = declared_functions()
    assert len(names) >= 16
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "liblsf_hip.so does not export %s" % name
        assert name in pkg._lib.PROTOTYPES, "no ctypes prototype for %s" % name
    assert sorted(pkg._lib.PROTOTYPES) == names
    assert pkg._lib.lib.lsf_abi_version() == pkg._lib.ABI_VERSION == 4
    assert pkg._lib.lib.lsf_target_arch() == b"gfx950"


def test_the_product_library_carries_what_the_product_runs(pkg):
    """the chain kernel (K iterations per launch, measured 4 % slower: DESIGN.md section 7, round 3) is a measurement tool
    under tools/chain/ with a library of its own -- neither the product library nor the package know it"""
    product = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in ("lsf_slavcheva_state_chain", "lsf_state_chain_plan", "lsf_state_chain_shape"):
        assert not hasattr(product, name), "the product library still carries %s" % name
    assert b"chain" not in open(pkg._lib.LIB_PATH, "rb").read()
    assert not os.path.exists(os.path.join(ROOT, "include", "lsf_hip_chain.h"))
    assert not hasattr(pkg._lib, "chain_lib") and not hasattr(pkg.device if hasattr(pkg, "device") else pkg._lib, "StateChain")


def test_few_environment_knobs_in_the_package(pkg):
    """measurement switches do not become product code paths: what the package reads from the environment is this list
    (every entry either has a test of both settings or names a file to load)"""
    allowed = {"LSF_HIP_LIBRARY",        # another build of the same library (A/B tools: tools/ab_state_kernel.py)
               "LSF_SLAB_TRANSPORT",     # rccl | torch: tests/slab_loopback_worker.py runs both
               "LSF_SLAB_FACES",         # compact | full: tests/slab_loopback_worker.py runs both
               "LSF_SPARSE_REACH",       # 0 = fully initialised states: tests/test_gpu_sparse_state.py, test_gpu_run_path.py
               "LSF_SPARSE_MIN_VOXELS"}  # tests force sparse states at small sizes
    found = set()
    package = os.path.join(ROOT, "levelsetfusion-python_amd")
    for folder, _, files in os.walk(package):
        for name in files:
            if name.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(folder, name)).read()
                found |= set(re.findall(r"environ(?:\.get)?\(\s*\"(LSF_[A-Z0-9_]+)\"", text))
                found |= set(re.findall(r"environ\[\s*\"(LSF_[A-Z0-9_]+)\"", text))
                found |= set(re.findall(r"getenv\(\s*\"(LSF_[A-Z0-9_]+)\"", text))
    assert found <= allowed, sorted(found - allowed)
    assert len(found) <= 10


def test_abi_hash_binds_library_binding_and_header(pkg, tmp_path):
    """ABI drift is refused, not remembered: the library carries the hash of the header it was compiled from, the
    binding the hash of the header it was written against, and both must be the header in the tree"""
    from levelsetfusion_python_amd import _build
    L = pkg._lib
    assert _build.abi_hash() == L.HEADER_ABI_HASH == L.lib.lsf_abi_hash().decode()
    # comments and layout do not count, a struct member or a prototype does
    text = open(HEADER).read()
    cosmetic = tmp_path / "cosmetic.h"
    cosmetic.write_text(text.replace("int lsf_abi_version(void);", "int   lsf_abi_version( void ) ;  /* same */\n")
                        .replace("( void ) ;", "(void);"))
    assert _build.abi_hash(str(cosmetic)) == _build.abi_hash()
    grown = tmp_path / "grown.h"
    grown.write_text(text.replace("int32_t packed_nz", "int32_t one_more_member;\n    int32_t packed_nz", 1))
    assert "one_more_member" in grown.read_text() and _build.abi_hash(str(grown)) != _build.abi_hash()
    changed = tmp_path / "changed.h"
    changed.write_text(text.replace("const int32_t *skip_flag,", "", 1))
    assert "skip_flag," not in changed.read_text() and _build.abi_hash(str(changed)) != _build.abi_hash()


def test_code_object_targets_gfx950_only(pkg):
    data = open(pkg._lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert other not in data


def test_ctypes_structs_match_header_layout(pkg):
    L = pkg._lib
    assert ctypes.sizeof(L.Grid) == 52 and L.Grid.energy_z_begin.offset == 32 and L.Grid.ny_global.offset == 40
    assert ctypes.sizeof(L.SlabLayoutC) == 32 and ctypes.sizeof(L.SlabPart) == 56 + 16 + 16 + 8 + 8
    assert ctypes.sizeof(L.IterationRecord) == 32768 == L.RECORD_BYTES and ctypes.sizeof(L.RecordSlot) == 4096
    from levelsetfusion_python_amd import slab
    assert (slab.RECORD_SLOTS, slab.SLOT_WORDS) == (L.RECORD_SLOTS, L.SLOT_WORDS)
    assert ctypes.sizeof(L.Gate) == 24 and L.Gate.mode.offset == 8 and L.Gate.a.offset == 12
    assert ctypes.sizeof(L.HierParams) == 40 and L.HierParams.packed_nz.offset == 32
    assert ctypes.sizeof(L.SlavchevaParams) == 56 and L.SlavchevaParams.rate.offset == 8
    text = open(HEADER).read()
    # lsf_run_loop, lsf_slab_run (round 6): the out members' offsets are what the binding reads back
    assert ctypes.sizeof(L.RunLoop) == 24 and L.RunLoop.check_interval.offset == 16
    assert ctypes.sizeof(L.StateRunResult) == 48 and L.StateRunResult.compact_faces.offset == 44
    assert L.SlabRun.layout.offset == ctypes.sizeof(L.StateRun) == 136 and L.SlabRun.cut_slices.offset == 176
    assert L.StateRun.box_all.offset == 128
    assert L.SlabRun.cut_entries.offset == 464 and L.SlabRun.out_face_entries.offset == 464 + 2 * 8 * L.SLAB_MAX_CUTS + 8
    assert ctypes.sizeof(L.SlabRun) == 1632
    for macro, value in (("LSF_ABI_VERSION", 4), ("LSF_SLAB_MAX_CUTS", L.SLAB_MAX_CUTS),
                         ("LSF_MAX_KERNEL_TAPS", L.MAX_KERNEL_TAPS),
                         ("LSF_SMOOTHING_KILLING", L.SMOOTHING_KILLING),
                         ("LSF_DATA_THRESHOLDED_FDM", L.DATA_THRESHOLDED_FDM),
                         ("LSF_ENERGY_VECTORIZED", L.ENERGY_VECTORIZED), ("LSF_GATE_SLAVCHEVA", L.GATE_SLAVCHEVA)):
        assert re.search(r"#define\s+%s\s+%d\b" % (macro, value), text), macro


def test_argument_errors_are_reported_not_launched(pkg):
    """host-side validation returns LSF_ERR_* before anything touches a device"""
    L = pkg._lib
    bad = L.Grid(4, 1, 8, 8, 0, 1, 0, 0)
    assert L.lib.lsf_warp_field(1, 1, 1, ctypes.byref(bad), 1.0, None) == -2          # dims
    g2 = L.Grid(2, 3, 8, 8, 0, 1, 0, 0)
    assert L.lib.lsf_warp_field(1, 1, 1, ctypes.byref(g2), 1.0, None) == -2           # 2-D needs nz == 1
    g3 = L.Grid(3, 4, 8, 8, 2, 9, 0, 0)
    assert L.lib.lsf_warp_field(1, 1, 1, ctypes.byref(g3), 1.0, None) == -1           # z range
    ok = L.Grid(3, 4, 8, 8, 0, 4, 0, 0)
    assert L.lib.lsf_warp_field(None, 1, 1, ctypes.byref(ok), 1.0, None) == -1        # null pointer
    taps = (ctypes.c_double * 40)()
    assert L.lib.lsf_convolve_axis(1, 2, None, ctypes.byref(ok), 3, 0, taps, 33, None, None) == -3
    assert L.lib.lsf_convolve_axis(1, 1, None, ctypes.byref(ok), 3, 0, taps, 7, None, None) == -1  # in place
    # the one-launch 3-D filter: whole arrays, nx % 4 == 0, 3 / 5 / 7 / 9 taps -- anything else is refused, not launched
    assert L.lib.lsf_convolve_xyz(1, 2, None, 0.0, ctypes.byref(ok), 3, taps, 4, None, None) == -3
    assert L.lib.lsf_convolve_xyz(1, 1, None, 0.0, ctypes.byref(ok), 3, taps, 7, None, None) == -1    # in place
    assert L.lib.lsf_convolve_xyz(1, 2, 2, 0.1, ctypes.byref(ok), 3, taps, 7, None, None) == -1       # warp aliases out
    bad_nx, flat = L.Grid(3, 4, 8, 6, 0, 4, 0, 0), L.Grid(2, 1, 8, 8, 0, 1, 0, 0)
    assert L.lib.lsf_convolve_xyz(1, 2, None, 0.0, ctypes.byref(bad_nx), 3, taps, 7, None, None) == -2
    assert L.lib.lsf_convolve_xyz(1, 2, None, 0.0, ctypes.byref(flat), 2, taps, 7, None, None) == -2
    assert L.lib.lsf_convolve_xyz(1, 2, None, 0.0, ctypes.byref(L.Grid(3, 4, 8, 8, 2, 2, 0, 0)), 3, taps, 7, None,
                                  None) == 0                                                      # empty z-range
    # the band-only finalize: whole arrays, at most two lists, statistics need canonical + scratch
    lists = (ctypes.c_void_p * 2)(1, 1)
    counts = (ctypes.c_int64 * 2)(5, -1)
    assert L.lib.lsf_state_finalize_listed(1, 1, 1, 1, ctypes.byref(ok), lists, counts, 3, 0, -1, 0.0, None, None,
                                           None, None, 0, 0.0, None) == -1                        # three lists
    assert L.lib.lsf_state_finalize_listed(1, 1, 1, 1, ctypes.byref(ok), lists, counts, 2, 0, -1, 0.0, None, None,
                                           None, None, 0, 0.0, None) == -1                        # negative count
    assert L.lib.lsf_state_finalize_listed(1, None, 1, 1, ctypes.byref(ok), lists, counts, 1, 0, -1, 0.0, 1, 1,
                                           None, None, 0, 0.0, None) == -1                        # statistics, no canonical
    assert L.lib.lsf_state_finalize_listed(1, 1, 1, 1, ctypes.byref(L.Grid(3, 4, 8, 8, 1, 4, 0, 0)), lists, counts, 1,
                                           0, -1, 0.0, None, None, None, None, 0, 0.0, None) == -1  # not a whole array
    with pytest.raises(pkg._lib.LsfHipError):
        pkg._lib.check(-2, "x")


def test_round6_entry_points_validate_on_the_host(pkg):
    """the entry points added in round 6 -- whole calls enqueued by the library (lsf_run_loop, lsf_slab_run_*,
    lsf_sobolev_run_finish), the blocked 2-D level, lsf_zero_listed4, lsf_slab_comm_info -- refuse bad arguments before
    anything touches a device (no GPU here)"""
    L = pkg._lib
    lib = L.lib
    g2, g3 = L.Grid(2, 1, 64, 64, 0, 1, 0, 0), L.Grid(3, 8, 32, 32, 0, 8, 0, 0)
    one, two, three, four, five = (ctypes.c_void_p(k * 4096) for k in (1, 2, 3, 4, 5))
    # lsf_hier_level_run_2d: 2-D, Tikhonov, update applied, no energies, 1..8 iterations per launch, distinct buffers
    ok = L.HierParams(1.0, 0.05, 0.1, 1, 1, 0)
    call = lib.lsf_hier_level_run_2d
    assert call(one, one, two, three, four, five, ctypes.byref(g3), ctypes.byref(ok), None, 0, one, 4, 8, 0.0, None) == -2
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(L.HierParams(1.0, 0.05, 0.1, 1, 0, 0)), None, 0, one, 4, 8, 0.0,
                None) == -2  # no gradient kernel: the iteration applies the update itself
    taps7 = (ctypes.c_double * 7)(*([0.1] * 7))
    filtered = L.HierParams(1.0, 0.05, 0.1, 1, 0, 0)
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(ok), taps7, 7, one, 4, 2, 0.0, None) == -2  # ... behind the filter with one
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(filtered), taps7, 7, one, 4, 3, 0.0, None) == -1  # 3 x 4 rings
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(filtered), taps7, 4, one, 4, 1, 0.0, None) == -2  # four taps
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(filtered), None, 7, one, 4, 2, 0.0, None) == -1  # taps missing
    assert call(one, one, two, two, four, five, ctypes.byref(g2), ctypes.byref(ok), None, 0, one, 4, 8, 0.0, None) == -1
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(ok), None, 0, one, 4, 0, 0.0, None) == -1
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(ok), None, 0, one, 4, 9, 0.0, None) == -1
    assert call(one, one, two, three, four, five, ctypes.byref(g2), ctypes.byref(ok), None, 0, one, 0, 8, 0.0, None) == 0  # nothing to do
    # lsf_convolve_xy: 3-D, nx % 4 == 0, 3 / 5 / 7 / 9 taps
    xy = lib.lsf_convolve_xy
    assert xy(one, one, ctypes.byref(g3), 3, (ctypes.c_double * 9)(), 7, None, None) == -1      # in == out
    assert xy(one, two, ctypes.byref(g2), 2, (ctypes.c_double * 9)(), 7, None, None) == -2      # 2-D
    assert xy(one, two, ctypes.byref(L.Grid(3, 8, 32, 30, 0, 8, 0, 0)), 3, (ctypes.c_double * 9)(), 7, None, None) == -2  # nx % 4
    assert xy(one, two, ctypes.byref(g3), 3, (ctypes.c_double * 9)(), 4, None, None) == -3      # four taps
    # lsf_convolve_axis_update: the last axis only, 3 / 5 / 7 / 9 taps, distinct buffers
    taps = (ctypes.c_double * 9)(*([0.1] * 9))
    upd = lib.lsf_convolve_axis_update
    assert upd(one, one, three, 0.1, ctypes.byref(g3), 3, 2, taps, 7, None, None) == -1           # in == out
    assert upd(one, two, two, 0.1, ctypes.byref(g3), 3, 2, taps, 7, None, None) == -1             # warp == out
    assert upd(one, two, None, 0.1, ctypes.byref(g3), 3, 2, taps, 7, None, None) == -1            # no warp
    assert upd(one, two, three, 0.1, ctypes.byref(g3), 3, 0, taps, 7, None, None) == -1           # the x pass
    assert upd(one, two, three, 0.1, ctypes.byref(g2), 2, 2, taps, 7, None, None) == -1           # axis 2 of a 2-D field
    assert upd(one, two, three, 0.1, ctypes.byref(g3), 3, 2, taps, 4, None, None) == -2           # four taps
    # lsf_zero_listed4
    assert lib.lsf_zero_listed4(None, ctypes.byref(g3), one, 4, 0, None) == -1
    assert lib.lsf_zero_listed4(one, ctypes.byref(g3), None, 4, 0, None) == -1
    assert lib.lsf_zero_listed4(one, ctypes.byref(g2), one, 4, 1, None) == -2                    # bricks are 3-D
    assert lib.lsf_zero_listed4(one, ctypes.byref(L.Grid(3, 8, 30, 32, 0, 8, 0, 0)), one, 4, 1, None) == -2
    assert lib.lsf_zero_listed4(one, ctypes.byref(g3), one, 0, 1, None) == 0                     # an empty list
    # the loop condition of lsf_state_run_finish / lsf_sobolev_run_finish
    run = L.StateRun()
    host = (ctypes.c_int64 * 8)()
    run.live, run.canonical, run.state[0], run.state[1] = 4096, 8192, 12288, 16384
    run.prepare_scratch, run.totals_device, run.totals_host, run.grid = 20480, 24576, ctypes.addressof(host), g3
    buf = (ctypes.c_int64 * 64)()
    res = L.StateRunResult(ctypes.addressof(buf), ctypes.addressof(buf), ctypes.addressof(buf), ctypes.addressof(buf))
    params = L.SlavchevaParams()
    finish = lib.lsf_state_run_finish

    def state_finish(iterations, loop):
        return finish(ctypes.byref(run), ctypes.byref(params), one, one, None, None, one, iterations,
                      ctypes.byref(loop) if loop is not None else None, one, 0.0, None, None, one, one, ctypes.byref(res), None)
    assert state_finish(10, L.RunLoop(0, 10, 0.1, 10.0, 4, 0)) == -1    # min_iterations 0: the caller's business
    assert state_finish(10, L.RunLoop(1, 10, 0.1, 10.0, 0, 0)) == -1    # check_interval
    assert state_finish(12, L.RunLoop(1, 10, 0.1, 10.0, 4, 0)) == -1    # records != max(min, max)
    taps = (ctypes.c_double * 7)(*([1.0 / 7] * 7))
    sob = lib.lsf_sobolev_run_finish

    def sobolev_finish(r, g_a=three, g_b=four):
        return sob(ctypes.byref(r), ctypes.byref(params), taps, 7, one, one, one, two, g_a, g_b, one, 5, None, one, 0.0, None,
                   None, one, one, ctypes.byref(res), None)
    assert sobolev_finish(run) == -1                                     # begun without box_scratch / box_all
    run.box_scratch, run.box_all = 28672, 1
    assert sobolev_finish(run, g_a=three, g_b=three) == -1               # one gradient buffer twice
    # lsf_slab_run_begin / _finish, lsf_slab_comm_info
    assert lib.lsf_slab_run_begin(None, None) == -1
    srun = L.SlabRun()
    assert lib.lsf_slab_run_begin(ctypes.byref(srun), None) == -1        # nothing filled in
    srun.base = run
    srun.base.box_scratch, srun.base.box_all = None, 0
    srun.base.grid = L.Grid(3, 12, 32, 32, 0, 12, 0, 0)
    srun.layout = L.SlabLayoutC(12, 32, 32, 2, 10, 2, 0, 0)
    srun.exchange_interval = 3                                           # neither 1 nor the halo
    assert lib.lsf_slab_run_begin(ctypes.byref(srun), None) == -1
    srun.exchange_interval = 2
    srun.layout = L.SlabLayoutC(12, 32, 30, 2, 10, 2, 0, 0)              # does not match the grid
    assert lib.lsf_slab_run_begin(ctypes.byref(srun), None) == -1
    assert lib.lsf_slab_run_finish(ctypes.byref(srun), None, ctypes.byref(params), one, one, one, None, one, 5, one, one, one,
                                   ctypes.byref(res), None) == -1        # no communicator
    rank, count = ctypes.c_int32(), ctypes.c_int32()
    assert lib.lsf_slab_comm_info(None, ctypes.byref(rank), ctypes.byref(count)) == -1


def test_slab_runtime_argument_errors(pkg):
    """the z-slab runtime (lsf_slab.hip) validates on the host before touching a device or RCCL"""
    L = pkg._lib
    layout = L.SlabLayoutC(16, 8, 8, 4, 12, 2, 0, 0)
    params = L.SlavchevaParams()
    grid = L.Grid(3, 16, 8, 8, 4, 12, 0, 0)
    # no communicator / null state for an exchanging call
    assert L.lib.lsf_slab_state_iteration(None, 1, 1, 2, ctypes.byref(layout), None, 0, None, 0, ctypes.byref(params),
                                          None, 1, L.SLAB_EXCHANGE, None, None) == -1
    # unknown mode
    assert L.lib.lsf_slab_state_iteration(None, 1, 1, 2, ctypes.byref(layout), None, 0, None, 0, ctypes.byref(params),
                                          None, 1, 7, None, None) == -1
    # launches only, but no record
    assert L.lib.lsf_slab_state_iteration(None, 1, 1, 2, ctypes.byref(layout), None, 0, None, 0, ctypes.byref(params),
                                          None, None, L.SLAB_LAUNCH, None, None) == -1
    # RESUME needs the communicator that holds the pending exchange
    assert L.lib.lsf_slab_state_iteration(None, 1, 1, 2, ctypes.byref(layout), None, 0, None, 0, ctypes.byref(params),
                                          None, 1, L.SLAB_RESUME, None, None) == -1
    handle = ctypes.c_void_p()
    ident = (ctypes.c_uint8 * 128)()
    assert L.lib.lsf_slab_comm_create(None, None, 0, 1, ctypes.byref(handle)) == -1              # no id
    assert L.lib.lsf_slab_comm_create(None, ctypes.cast(ident, ctypes.c_void_p), 3, 2, ctypes.byref(handle)) == -1  # rank
    assert L.lib.lsf_slab_unique_id(None, None) == -1
    assert L.lib.lsf_slab_comm_destroy(None) == 0
    # the face counts' collective: a communicator, four counts in, a table out
    four = (ctypes.c_int64 * 4)(1, 2, 3, 4)
    assert L.lib.lsf_slab_face_counts_begin(None, four) == -1
    assert L.lib.lsf_slab_face_counts_end(None, four) == -1
    # one 1024-voxel chunk: two prefix arrays, ballots, the unlisted counts, two verdicts, the scan's per-tile totals (4 ints
    # per tile of 4096 chunks, + 1 tile)
    assert L.lib.lsf_state_prepare_scratch_elements(ctypes.byref(grid)) == 2 * 2 + 64 * 1 + 2 + 2 + 2 + 4 * 2
    # the sparse initialisation: whole arrays, a reach of 1..8 voxels, at least one state
    whole = L.Grid(3, 16, 8, 8, 0, 16, 0, 0)
    assert L.lib.lsf_state_pack_needed(1, 1, 1, ctypes.byref(grid), 1, 2, 0, None) == -1      # not a whole array
    assert L.lib.lsf_state_pack_needed(1, 1, 1, ctypes.byref(whole), 1, 0, 0, None) == -1     # reach
    assert L.lib.lsf_state_pack_needed(1, None, None, ctypes.byref(whole), 1, 2, 0, None) == -1

    assert L.lib.lsf_band_list_fill_prepared(ctypes.byref(grid), L.BAND_INTERIOR, 1, 1, None) == -1  # not a whole array


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = np.zeros((16, 16), np.float32)
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        pkg.HierarchicalOptimizer2d(maximum_chunk_size=4).optimize(c, c)
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        pkg.SlavchevaOptimizer2d(out_path=None, field_size=16).optimize(c.copy(), c)
    # nothing under the package imports the oracle
    pkg_dir = os.path.dirname(pkg._lib.__file__)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "lsf_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_pyramid_rules(pkg):
    from levelsetfusion_python_amd.engine_common import _conv_axis_order, pyramid_level_count
    assert pyramid_level_count((128, 128), 8) == 4 and pyramid_level_count((16, 16), 4) == 3
    assert pyramid_level_count((64, 64, 64), 8) == 4
    for shape, chunk in (((12, 16), 4), ((16, 16), 3), ((8, 8), 8), ((16, 4), 4)):
        with pytest.raises(ValueError):
            pyramid_level_count(shape, chunk)
    assert _conv_axis_order(2) == [1, 0] and _conv_axis_order(3) == [0, 1, 2]


def test_flag_folding_and_attributes(pkg):
    o = pkg.HierarchicalOptimizer2d(tikhonov_term_enabled=True, tikhonov_strength=0.0, gradient_kernel_enabled=True,
                                    kernel=None)
    assert not o.tikhonov_term_enabled and not o.gradient_kernel_enabled
    o = pkg.HierarchicalOptimizer2d(tikhonov_term_enabled=False, tikhonov_strength=0.3,
                                    kernel=np.ones(3), gradient_kernel_enabled=False)
    assert o.tikhonov_strength == 0.0 and o.gradient_kernel is None
    o = pkg.HierarchicalOptimizer2d(kernel=np.array([0.1, 0.8, 0.1]))
    assert o.tikhonov_term_enabled and o.gradient_kernel_enabled and o.rate == 0.1
    assert o.maximum_iteration_count == 100 and o.maximum_warp_update_threshold == 0.001
    vp = pkg.HierarchicalOptimizer2d.VerbosityParameters(print_max_warp_update=True)
    assert vp.print_per_iteration_info and vp.print_per_level_info
    assert pkg.ComputeMethod.DIRECT.value == 0 and pkg.ComputeMethod.VECTORIZED.value == 1
    assert [m.name for m in pkg.DataTermMethod] == ["BASIC", "BASIC_CPP", "THRESHOLDED_FDM"]
    assert [m.name for m in pkg.SmoothingTermMethod] == ["TIKHONOV", "KILLING"]
    with pytest.raises(ValueError):
        pkg.SlavchevaOptimizer2d(out_path=None, sobolev_smoothing_enabled=True, sobolev_kernel=None)


def test_sobolev_filter_host_code(pkg, ref_leaf):
    for s, lam, k in ((3, 0.1, "k3"), (7, 0.1, "k7"), (9, 0.15, "k9")):
        got = pkg.generate_1d_sobolev_kernel(s, lam)
        assert got.dtype == np.float32 and np.abs(got - ref_leaf["sobolev." + k]).max() <= 1e-7
    from levelsetfusion_python_amd.math_utils.convolution import sobolev_kernel_1d
    assert np.abs(sobolev_kernel_1d - ref_leaf["sobolev.hardcoded7"]).max() == 0.0


def test_synthetic_depth_frames_are_the_oracle_s(pkg):
    """the package's closed-form depth frames (bench.py --data depth) are the ones the oracle / the TSDF tests use"""
    from levelsetfusion_python_amd import synthetic
    from oracle import lsf_oracle as O
    for kw in (dict(), dict(shift_px=2.0, nearer_m=0.008)):
        d = synthetic.depth_image(**kw)
        assert d.dtype == np.uint16 and d.shape == (480, 640) and np.array_equal(d, O.synthetic_depth_image(**kw))


def test_record_decoding(pkg):
    """a record is 8 partial slots (lsf_iteration_record): value = max of the packed maxima, sum of the energies"""
    from levelsetfusion_python_amd import device as dev

    def pack(val, idx):
        p = (np.uint64(np.float32(val).view(np.uint32)) << np.uint64(32)) | np.uint64((~np.uint32(idx)) & 0xFFFFFFFF)
        return np.array([p], np.uint64).view(np.int64)[0]
    raw = np.zeros((3, dev.RECORD_WORDS), np.int64)
    slots = dev.slot_view(raw)
    slots[0, 2, 0] = pack(0.125, 77)
    slots[0, 5, 0] = pack(0.125, 12)       # same length, smaller index: numpy's first arg-max
    slots[0, 7, 0] = pack(0.0625, 3)
    slots[0, 2, 1:4] = np.array([1.5, 2.5, 3.5]).view(np.int64)
    slots[0, 6, 1:4] = np.array([0.25, 0.5, 1.0]).view(np.int64)
    d = dev.decode_records(raw)
    assert list(d["executed"]) == [True, False, False]
    assert d["max_value"][0] == np.float32(0.125) and d["argmax"][0] == 12
    assert (d["data_energy"][0], d["smoothing_energy"][0], d["level_set_energy"][0]) == (1.75, 3.0, 4.5)
    # contiguous records go through the library (lsf_records_decode, a host function), anything else through numpy: the same
    # values from both, on random records, for 8 slots (one rank) and 24 (three ranks' slots side by side), 4 and 512 words
    rng = np.random.default_rng(3)
    for n_slots, words in ((8, 4), (24, 4), (8, 512)):
        wide = np.zeros((7, n_slots, words + 2), np.int64)
        wide[:, :, 0] = [[pack(v, i) for v, i in zip(rng.random(n_slots) * (rng.random(n_slots) > 0.3),
                                                     rng.integers(0, 2 ** 31, n_slots))] for _ in range(7)]
        wide[:, :, 1:4] = rng.standard_normal((7, n_slots, 3)).view(np.int64)
        wide[3] = 0  # a record that was never executed
        through_numpy = dev.decode_records(wide[:, :, :words])       # a strided view: not contiguous
        through_library = dev.decode_records(np.ascontiguousarray(wide[:, :, :words]))
        assert set(through_numpy) == set(through_library)
        for key in through_numpy:
            a, b = through_numpy[key], through_library[key]
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), key
        assert not through_library["executed"][3] and through_library["executed"][0]
    assert pkg._lib.lib.lsf_records_decode(None, 1, 8, 4, None, None, None, None) == -1
    assert pkg._lib.lib.lsf_records_decode(1, 1, 8, 3, 1, 1, 1, 1) == -1  # a slot has at least its four used words


def test_slab_layout(pkg):
    from levelsetfusion_python_amd.slab import SlabLayout
    a, b, c = (SlabLayout(24, r, 3, 2) for r in range(3))
    assert (a.z0, a.z1, a.halo_lo, a.halo_hi, a.nz_local, a.z_begin, a.z_end, a.z_global_offset) == \
        (0, 8, 0, 2, 10, 0, 8, 0)
    assert (b.z0, b.z1, b.halo_lo, b.halo_hi, b.nz_local, b.z_begin, b.z_end, b.z_global_offset) == \
        (8, 16, 2, 2, 12, 2, 10, 6)
    assert (c.nz_local, c.z_begin, c.z_end, c.z_global_offset) == (10, 2, 10, 14)
    assert b.local_slice() == slice(6, 18) and c.owned_local() == slice(2, 10)
    one = SlabLayout(16)
    assert one.nz_local == 16 and one.halo_lo == one.halo_hi == 0
    with pytest.raises(ValueError):
        SlabLayout(10, 0, 3, 1)
    with pytest.raises(ValueError):
        SlabLayout(8, 0, 4, 3)


def test_multipair_report_tables(pkg, tmp_path):
    """host side of the multi-pair driver: table layout (2 + 17 columns per level), analysis and case files"""
    from levelsetfusion_python_amd.convergence_report import (ConvergenceReport, Location, TsdfDifferenceStatistics,
                                                              WarpDeltaStatistics)
    from levelsetfusion_python_amd.experiment import multipair as mp
    loc = Location(3, 4)
    assert loc == (3, 4) and loc.x == 3 and loc.y == 4 and Location((1, 2, 5)).z == 5

    def report(count, reached):
        return ConvergenceReport(count, reached, WarpDeltaStatistics(0.5, 0.0, 0.2, 0.1, 0.05, (3, 4), False, False),
                                 TsdfDifferenceStatistics(0.0, 0.3, 0.1, 0.05, (5, 6)))
    sets = [[report(10, False), report(100, True)], [report(20, False), report(50, False)]]
    df = mp.post_process_convergence_report_sets(sets, [(7, 200), (8, 200)])
    assert len(df.columns) == 2 + 17 * 2 and mp.infer_level_count(df) == 2
    assert list(df.columns[:3]) == ["canonical_frame", "pixel_row", "l0_iter_count"]
    assert list(df["l1_warp_delta_max_x"]) == [3, 3] and list(df["l1_diff_max_y"]) == [6, 6]
    assert mp.get_converged_ratio_for_level(df, 0) == 1.0 and mp.get_converged_ratio_for_level(df, 1) == 0.5
    assert mp.get_mean_iteration_count_for_level(df, 1) == 75.0
    text = mp.analyze_convergence_data(df, str(tmp_path))
    assert "level 1: 50.00%" in text and "level 0: 15.00" in text
    assert len(mp.save_bad_cases(df, str(tmp_path))) == 1
    mp.save_all_cases(df, str(tmp_path))
    assert open(str(tmp_path / "bad_cases.csv")).read().strip() == "7,200,3,4"
    assert len(open(str(tmp_path / "all_cases.csv")).read().strip().splitlines()) == 2
    c = np.zeros((4, 4), np.float32)
    mp.save_pair(str(tmp_path / "pairs"), 12, 300, c, c + 1)
    mp.save_pair(str(tmp_path / "pairs"), 3, 300, c, c + 2)
    pairs = mp.load_pairs(str(tmp_path / "pairs"))
    assert [(p[0], p[1]) for p in pairs] == [(3, 300), (12, 300)] and float(pairs[1][3][0, 0]) == 1.0


def test_bench_plain_form_starts_its_own_launcher(monkeypatch):
    """`python bench.py --gpus N` without torch.distributed.run in front: bench.main() hands the same arguments to a child
    `python -m torch.distributed.run --nproc-per-node N ... bench.py`, before anything touches the GPU, and exits with the
    child's status (the GPU suite runs the real thing: test_gpu_bench_contract.py)"""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    monkeypatch.setattr(bench.torch.cuda, "set_device", lambda *_: (_ for _ in ()).throw(AssertionError("GPU touched")))
    with pytest.raises(SystemExit) as exit_info:
        bench.main()
    assert exit_info.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_scaling_modes_parse():
    """bench.py's N > 1 modes exist and default as documented (no GPU needed to parse)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    old = sys.argv
    try:
        spec.loader.exec_module(bench)
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.scaling, a.pattern, a.halo, a.size, a.workload) == ("weak", None, None, 256, "killing")
        sys.argv = ["bench.py", "--gpus", "8", "--scaling", "strong", "--workload", "hier2d"]
        a = bench.parse()
        assert (a.scaling, a.gpus, a.size) == ("strong", 8, 512)
        assert [s[0] for s in bench.SECONDARY] == ["killing", "killing-pairs", "killing-default", "hier-tik", "hier-full", "config3",
                                                  "multiframe", "sobolev", "hier2d"]
    finally:
        sys.argv = old
