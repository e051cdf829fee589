#!/usr/bin/env python3
"""host time of a library-enqueued 256^3 call outside its two foreign calls: optimize() entry -> lsf_state_run_begin,
begin -> finish, finish's return -> optimize()'s return, and the bench loop's share (the copy that restores the live field).
The card idles through the first, the third and the fourth (tools/step_gaps.py shows them as gaps).  usage: host_gaps_run.py [size]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd import _lib  # noqa: E402
from levelsetfusion_python_amd.hostloop import parked_collector  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
canonical, live0 = sphere_pair(n, 3, "cuda")
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING, maximum_warp_length_lower_threshold=0.0,
                               max_iterations=50, min_iterations=50, check_interval=50)
marks = {}


class Timed:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in ("lsf_state_run_begin", "lsf_state_run_finish"):
            def inner(*a):
                t0 = time.perf_counter()
                r = fn(*a)
                marks[name] = (t0, time.perf_counter())
                return r
            return inner
        return fn


_lib.lib = Timed(_lib.lib)
import levelsetfusion_python_amd.engine_run as er  # noqa: E402
er._lib.lib = _lib.lib
live = torch.empty_like(live0)
rows = []
with parked_collector():
    for k in range(60):
        t_loop = time.perf_counter()
        live.copy_(live0)
        t_in = time.perf_counter()
        opt.optimize(live, canonical)
        t_out = time.perf_counter()
        b, f = marks["lsf_state_run_begin"], marks["lsf_state_run_finish"]
        rows.append((t_in - t_loop, b[0] - t_in, b[1] - b[0], f[0] - b[1], f[1] - f[0], t_out - f[1]))
r = np.array(rows[10:]) * 1e6
print("%d^3, microseconds (median of %d calls): restore copy launch %.1f | entry -> begin %.1f | begin %.1f | begin -> finish %.1f | "
      "finish %.1f | finish -> return %.1f | whole step %.1f" % ((n, len(r)) + tuple(np.median(r, axis=0)) + (np.median(r.sum(axis=1)),)))
