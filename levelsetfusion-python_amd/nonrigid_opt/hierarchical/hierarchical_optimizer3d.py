"""HierarchicalOptimizer3d -- the 3-D optimizer the reference only has in its un-vendored C++ extension
(cpp.HierarchicalOptimizer3d; constructor keywords as used by run_hierarchical_optimizer3d.py:81-98 and
run_hierarchical_optimizer3d_multipair.py:367-389).  optimize(canonical_field, live_field) -> warp (D, H, W, 3).
The 3-D arithmetic is the dimension generalisation of the 2-D Python path written down in DESIGN.md section 3."""
from enum import Enum

from .hierarchical_optimizer2d import _HierarchicalOptimizerBase


class HierarchicalOptimizer3d(_HierarchicalOptimizerBase):
    DIMS = 3

    class ResamplingStrategy(Enum):
        NEAREST_AND_AVERAGE = 0   # 2x2x2 mean restrict, repeat prolong (the Python reference's only strategy)
        LINEAR = 1                # math_utils/resampling.py: 4x4x4 [1,3,3,1]/8 restrict, 0.75/0.25 trilinear prolong

    def __init__(self, *args, resampling_strategy=None, **kwargs):
        strategy = resampling_strategy or HierarchicalOptimizer3d.ResamplingStrategy.NEAREST_AND_AVERAGE
        if not isinstance(strategy, HierarchicalOptimizer3d.ResamplingStrategy):
            raise ValueError("resampling_strategy must be a HierarchicalOptimizer3d.ResamplingStrategy")
        super().__init__(*args, linear_resampling=strategy == HierarchicalOptimizer3d.ResamplingStrategy.LINEAR,
                         **kwargs)
        self.resampling_strategy = strategy
