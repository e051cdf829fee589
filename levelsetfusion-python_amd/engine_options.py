"""Engine knobs that are not part of the reference's constructor signatures: how a call is walked, enqueued and
initialised.  Results never depend on them (tests hold the settings against each other); defaults are what measured
fastest.  The drop-in optimizers take them as `engine_options=dict(...)`; what a call actually
used is reported in `optimizer.engine.last_call`."""
import os
import types

# SlavchevaEngine
SLAVCHEVA_DEFAULTS = dict(
    use_band_list=True,    # False: the fused kernel walks every voxel (measurements, tests)
    library_run=True,      # whole calls enqueued by the library (lsf_state_run_* / lsf_slab_run_*); False: one foreign call
                           # per launch (tests hold the two against each other)
    box_walk=None,         # None: by band size (box_walk_min_band_bytes); True / False: always / never
    sobolev_boxes=True,    # SobolevFusion on whole 3-D volumes: y pass, z pass and update box by box (False: lists)
    # The ping-pong states are initialised only where an iteration can read them while every update stays below this many
    # voxels (0 = everywhere, as before round 4), for volumes of at least sparse_min_voxels (below, a call is launch-bound
    # and the initialisation passes cost nothing next to it).  None: LSF_SPARSE_REACH / LSF_SPARSE_MIN_VOXELS, else 2 / 2^21
    sparse_reach=None,
    sparse_min_voxels=None,
    # The library-enqueued call walks the INTERIOR band voxels box by box (lsf_slavcheva_state_iteration_boxes:
    # neighbourhoods staged through LDS) instead of entry by entry when the listed voxels of the two ping-pong states -- 32
    # bytes each -- and what else an iteration touches crowd the 256 MB Infinity Cache: the list walk's 18 loads per voxel
    # then miss to HBM and its L1s stand at their in-flight limit (profiles/r05_pmc_l2_tcp.txt), the box walk's four
    # coalesced loads per 64 voxels do not: 120-125 against 138-144 us per 512^3 launch.  Measured on one box, list / box
    # walk in us per launch (tools/box_kernel_ab.py, profiles/r05_box_walk_by_size.txt): sphere pairs 320^3 (84 MB of listed
    # states) 43.6 / 43.6, 384^3 (122 MB) 65.2 / 62.2, 448^3 (169 MB) 103.1 / 88.8, 512^3 (224 MB) 138-144 / 119-125; the
    # 512^3 depth pair (119 MB) 76.0 / 66.9.  Below ~100 MB the two are level (256^3: 30.8 us both) and the list walk needs
    # no boxes built.
    box_walk_min_band_bytes=100 * 1000 * 1000,
    box_walk_min_voxels=1 << 25,  # (volumes below this never reach the band size above: the boxes are not even counted)
)

# HierarchicalEngine
HIERARCHICAL_DEFAULTS = dict(
    use_graphs=True,               # HIP-graph replay for launch-bound levels
    graph_max_voxels=1 << 21,      # ... i.e. levels of at most this many voxels
    # 3-D levels from 2^23 voxels up: lsf_convolve_xyz instead of three passes (0.21 against 0.25 ms at 256^3, 1.29 against
    # 1.9 ms at 512^3; below that its 64 x 16-column blocks are too few to fill the GPU: 0.045 / 0.035 ms at 128^3)
    fused_filter=True,             # (False: three convolve_axis passes -- measurements, tests)
    fused_filter_min_voxels=1 << 23,
    defer_maximum=True,            # (False: every iteration keeps its own maximum pass)
    blocked_levels=True,           # 2-D levels (Tikhonov term, +/- gradient kernel): K iterations per launch inside LDS tiles, a stop test
                                   # looked at launch by launch (lsf_hier_level_run_2d); False: one launch per iteration
)


def apply(engine, defaults, options):
    """set every knob of `defaults` on the engine, overridden by `options` (a dict or None); unknown names are refused"""
    options = dict(options or {})
    unknown = sorted(set(options) - set(defaults))
    if unknown:
        raise TypeError("unknown engine option(s) %s (known: %s)" % (", ".join(unknown), ", ".join(sorted(defaults))))
    for name, value in defaults.items():
        setattr(engine, name, options.get(name, value))
    if "sparse_reach" in defaults:
        if engine.sparse_reach is None:
            engine.sparse_reach = int(os.environ.get("LSF_SPARSE_REACH", "2"))
        if engine.sparse_min_voxels is None:
            engine.sparse_min_voxels = int(os.environ.get("LSF_SPARSE_MIN_VOXELS", str(1 << 21)))


def new_call_report():
    """what the last optimize() call took: tests and measurements read it instead of private attributes"""
    return types.SimpleNamespace(sparse_states=False, box_walk=False, sobolev_boxes=False, library_run=False,
                                 blocked_levels=0)
