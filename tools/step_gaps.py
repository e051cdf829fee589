"""The card's side of ONE bench step, from a rocprofv3 --kernel-trace of bench.py: every kernel of the last complete step
(from one counting pass to the next) with its start relative to the step's first kernel, its duration and the idle gap
in front of it; iteration launches are folded into one line.  usage: step_gaps.py <dir with *_kernel_trace.csv> [step from the end]"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|void |lsf::", "", r["Kernel_Name"]).split("(")[0][:58]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
starts = [i for i, r in enumerate(rows) if "state_prepare" in r[2] or "prepare_count" in r[2] or "band_count" in r[2]]
starts = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 10]
a, b = starts[-back - 1], starts[-back]
while a > 0 and rows[a][0] - rows[a - 1][1] < 60000 and "slavcheva_state" not in rows[a - 1][2] and a > starts[-back - 2] + 1 and "finalize" not in rows[a - 1][2] and "records_used" not in rows[a - 1][2]:
    a -= 1  # launches in front of the counting pass that belong to the step (copies, fills)
step = rows[a:b]
t0 = step[0][0]
print("step of %d kernels, %.1f us from its first kernel's start to the next step's first kernel" % (len(step), (rows[b][0] - t0) / 1e3))
prev_end = t0
it_n, it_busy, it_gap, it_first = 0, 0, 0, None
busy = 0
for s, e, name in step:
    gap, dur = s - prev_end, e - s
    busy += dur
    if "slavcheva_state_kernel" in name or "slavcheva_state_box_kernel" in name:
        if it_n == 0:
            it_first = (s - t0, gap)
        it_n += 1
        it_busy += dur
        it_gap += gap if it_n > 1 else 0
    else:
        if it_n:
            print("  +%8.1f us  gap %6.1f  %4d x slavcheva_state_kernel: busy %.1f us (mean %.2f), gaps between them %.1f us (mean %.2f)"
                  % (it_first[0] / 1e3, it_first[1] / 1e3, it_n, it_busy / 1e3, it_busy / 1e3 / it_n, it_gap / 1e3, it_gap / 1e3 / max(it_n - 1, 1)))
            it_n = it_busy = it_gap = 0
        print("  +%8.1f us  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap / 1e3, dur / 1e3, name))
    prev_end = e
if it_n:
    print("  +%8.1f us  gap %6.1f  %4d x slavcheva_state_kernel: busy %.1f us" % (it_first[0] / 1e3, it_first[1] / 1e3, it_n, it_busy / 1e3))
print("  busy %.1f us; idle from the last kernel's end to the next step's first kernel: %.1f us" % (busy / 1e3, (rows[b][0] - prev_end) / 1e3))
