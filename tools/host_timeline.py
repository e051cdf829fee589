#!/usr/bin/env python3
"""wall-clock timeline of the host side of one SlavchevaOptimizer3d.optimize() call (bench workload): start / end of the
main phases relative to the start of the call, averaged over a number of steps.  Shows where the GPU waits for the
host (work after a synchronising read, before the next launch).  Usage: host_timeline.py [size] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetfusion_python_amd as lsf  # noqa: E402
from levelsetfusion_python_amd import device as dev, engine  # noqa: E402
from levelsetfusion_python_amd.synthetic import sphere_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
marks = []


def wrap(owner, name, label=None):
    fn = getattr(owner, name)

    def inner(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            marks.append((label or name, t0, time.perf_counter()))
    setattr(owner, name, inner)


for owner, name in ((dev.StatePrepare, "__init__"), (dev.StatePrepare, "collect"), (dev, "new_records"),
                    (dev.IterationLauncher, "__init__"),
                    (engine.SlavchevaEngine, "_enqueue_state_iteration"), (engine.SlavchevaOutcome, "enqueue_finalize"),
                    (dev, "records_to_host"), (dev, "decode_records"), (engine.SlavchevaOutcome, "finalize"),
                    (engine.SlavchevaEngine, "optimize")):
    if hasattr(owner, name):
        wrap(owner, name, "%s.%s" % (getattr(owner, "__name__", owner), name))

canonical, live0 = sphere_pair(n, 3, torch.device("cuda", 0))
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                               maximum_warp_length_lower_threshold=0.0, max_iterations=50, min_iterations=50,
                               check_interval=50)
live = torch.empty_like(live0)
import gc  # noqa: E402
gc.disable()
acc = {}
total = 0.0
for s in range(steps + 5):
    torch.cuda.synchronize()
    del marks[:]
    t0 = time.perf_counter()
    live.copy_(live0)
    opt.optimize(live, canonical)
    t_ret = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if s < 5:
        continue
    total += t1 - t0
    seen = {}
    for label, a, b in marks:
        k = seen.get(label, 0)
        seen[label] = k + 1
        if label.endswith("_enqueue_state_iteration"):
            key = label + (" first" if k == 0 else " rest")
        else:
            key = "%s#%d" % (label, k)
        e = acc.setdefault(key, [0.0, 0.0, 0.0, 0])
        e[0] += a - t0 if e[3] % 49 == 0 or not key.endswith("rest") else 0.0
        e[1] += b - t0
        e[2] += b - a
        e[3] += 1
    e = acc.setdefault("optimize() returned", [0.0, 0.0, 0.0, 0])
    e[1] += t_ret - t0
    e[3] += 1
print("step %.3f ms over %d steps" % (total / steps * 1e3, steps))
print("%-58s %10s %10s %10s" % ("phase", "start us", "end us", "busy us"))
for key, (a, b, d, cnt) in sorted(acc.items(), key=lambda kv: kv[1][1] / max(kv[1][3], 1)):
    per = cnt / steps
    if key.endswith("rest"):
        print("%-58s %10s %10.1f %10.1f  (%d calls/step; end = mean end)" % (key, "", b / cnt * 1e6, d / steps * 1e6, per))
    else:
        print("%-58s %10.1f %10.1f %10.1f" % (key, a / cnt * 1e6, b / cnt * 1e6, d / cnt * 1e6))
