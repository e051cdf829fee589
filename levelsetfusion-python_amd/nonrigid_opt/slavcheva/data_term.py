"""Data-term method selector (mirrors nonrigid_opt/slavcheva/data_term.py:44-47 of the reference).
The arithmetic itself runs inside the fused HIP kernel (csrc/lsf_slavcheva.hip, voxel_gradient)."""
from enum import Enum


class DataTermMethod(Enum):
    BASIC = 0
    BASIC_CPP = 1          # the reference's C++ twin of BASIC; identical arithmetic here
    THRESHOLDED_FDM = 2    # the threshold picks the finite-difference direction (data_term.py:190-227)
