"""The card's side of ONE slab optimize() call, from a rocprofv3 --kernel-trace of tools/slab_nccl_loopback.py: every kernel of the
call (counting pass ... record words) with its start relative to the call's first kernel, its duration, the queue (stream) it
ran on and the gap to the previous kernel's end on ANY queue.  usage: slab_trace_call.py <dir with *_kernel_trace.csv> [which]
which: index among the calls that contain RCCL kernels, from the end (default 2 = the last library-enqueued slab call when the
tool alternates library / per-iteration / single)"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|void |lsf::", "", r["Kernel_Name"]).split("(")[0][:48]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
starts = [i for i, r in enumerate(rows) if "state_prepare" in r[2]]
calls = []
for k, a in enumerate(starts):
    b = starts[k + 1] if k + 1 < len(starts) else len(rows)
    seg = rows[a:b]
    if any("ncclDevKernel" in r[2] or "face_gather" in r[2] for r in seg):
        calls.append(seg)
seg = calls[-which]
# cut the segment at the record words / finalize of this call
end = max(i for i, r in enumerate(seg) if "record_words" in r[2] or "records_used" in r[2] or "finalize" in r[2]) + 1
seg = seg[:end]
t0 = seg[0][0]
print("slab call of %d kernels, %.1f us from its first kernel's start to its last kernel's end" % (len(seg), (seg[-1][1] - t0) / 1e3))
prev_end = t0
for s, e, name, q in seg:
    print("  +%8.1f us  gap %6.1f  dur %6.1f  q%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, name))
    prev_end = max(prev_end, e)
