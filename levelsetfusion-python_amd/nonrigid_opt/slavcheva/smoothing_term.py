"""Smoothing-term method selector (mirrors nonrigid_opt/slavcheva/smoothing_term.py:27-29 of the reference).
Tikhonov = -Laplacian of the previous update; Killing = approximately-Killing-vector-field regulariser.
Both run inside the HIP kernels of csrc/lsf_slavcheva.hip."""
from enum import Enum


class SmoothingTermMethod(Enum):
    TIKHONOV = 0
    KILLING = 1
