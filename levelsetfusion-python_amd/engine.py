"""Device-resident optimizer engines (dimension-generic, slab-aware).  The drop-in classes under nonrigid_opt/ are thin
shells around these.

All state lives in torch ROCm tensors; every per-voxel operation is a hand-written HIP kernel reached through the C ABI
(device.py -> liblsf_hip.so).  The host only enqueues launches -- or hands a whole call to the library -- and reads back
the tiny iteration records to learn whether the device-side convergence gate has closed.

This module is the import point; the code lives in
    engine_common.py     input conversion, pyramid level rule, graph-capture lock
    engine_options.py    the knobs that are not part of the reference's signatures, and the last call's report
    engine_hier.py       HierarchicalEngine (whole volumes and z-slabs, HIP graphs, persistent 2-D levels)
    engine_slavcheva.py  SlavchevaEngine: construction, the general call path, hooks, gradient_field
    engine_run.py        ... its library-enqueued calls on whole volumes (fixed-count, threshold-terminated)
    engine_slab.py       ... on z- / y-slabs: launch plan, exchange groups, compact faces, the wider re-run
    engine_sobolev.py    ... the SobolevFusion launch plans (float4 lists / boxes, planar fields)
    engine_outcome.py    what a call leaves behind (final fields on demand, the per-iteration log)
"""
from .engine_common import as_device_field, pyramid_level_count  # noqa: F401
from .engine_hier import HierarchicalEngine, LevelResult  # noqa: F401
from .engine_outcome import SlavchevaOutcome  # noqa: F401
from .engine_slavcheva import SlavchevaEngine  # noqa: F401
