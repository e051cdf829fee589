"""SlavchevaOptimizer3d -- the volumetric KillingFusion / SobolevFusion optimizer (BASELINE config 4 / 5).
The reference has no 3-D Slavcheva optimizer in Python; this is the dimension generalisation of
SlavchevaOptimizer2d (rules in DESIGN.md section 3; z-constant volumes reproduce the 2-D result bit for bit on
interior slices).  Same constructor keywords and optimize(live_field, canonical_field) convention; fields are
[z][y][x] float32, warps (D, H, W, 3).  Pass comm=SlabComm(...) to run one z-slab of a larger volume per GPU."""
from .slavcheva_optimizer2d import _SlavchevaOptimizerBase


class SlavchevaOptimizer3d(_SlavchevaOptimizerBase):
    DIMS = 3

    def __init__(self, out_path=None, *args, **kwargs):
        super().__init__(out_path, *args, **kwargs)
