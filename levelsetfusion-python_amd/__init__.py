"""levelsetfusion-python_amd -- MI355X-native non-rigid level-set (KillingFusion / SobolevFusion style) optimizers.

Drop-in for the numpy path of Algomorph/LevelSetFusion-Python's warp-field gradient descent:
    SlavchevaOptimizer2d(...).optimize(live_field, canonical_field)
    HierarchicalOptimizer2d(...).optimize(canonical_field, live_field)
plus their 3-D generalisations.  Host code is Python; device buffers are PyTorch-ROCm tensors; every per-voxel
operation is a hand-written HIP kernel (gfx950) behind the C ABI in include/lsf_hip.h.  There is no CPU
execution path: importing the package without liblsf_hip.so raises.

The directory name contains a hyphen, so import it through the loader module `levelsetfusion_python_amd`
at the repository root.
"""
from . import _lib  # noqa: F401  (raises ImportError loudly when the HIP library is missing)
from .nonrigid_opt.hierarchical.hierarchical_optimizer2d import HierarchicalOptimizer2d
from .nonrigid_opt.hierarchical.hierarchical_optimizer3d import HierarchicalOptimizer3d
from .nonrigid_opt.slavcheva.slavcheva_optimizer2d import (AdaptiveLearningRateMethod, ComputeMethod,
                                                           SlavchevaOptimizer2d)
from .nonrigid_opt.slavcheva.slavcheva_optimizer3d import SlavchevaOptimizer3d
from .nonrigid_opt.slavcheva.data_term import DataTermMethod
from .nonrigid_opt.slavcheva.smoothing_term import SmoothingTermMethod
from .nonrigid_opt.slavcheva.sobolev_filter import generate_1d_sobolev_kernel

__all__ = ["HierarchicalOptimizer2d", "HierarchicalOptimizer3d", "SlavchevaOptimizer2d", "SlavchevaOptimizer3d",
           "ComputeMethod", "AdaptiveLearningRateMethod", "DataTermMethod", "SmoothingTermMethod",
           "generate_1d_sobolev_kernel"]
