"""per-kernel totals of a rocprofv3 --kernel-trace run (csv): calls, total ms, mean us -- and the sum over all kernels.
usage: trace_totals.py <dir with *_kernel_trace.csv> [top]"""
import collections
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
acc = collections.defaultdict(list)
t_min, t_max = None, None
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:60]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    acc[name].append(b - a)
    t_min = a if t_min is None else min(t_min, a)
    t_max = b if t_max is None else max(t_max, b)
total = sum(sum(v) for v in acc.values())
print("kernels: %.2f ms busy over a %.2f ms span" % (total / 1e6, (t_max - t_min) / 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-62s n %5d total %8.2f ms mean %8.1f us %5.1f%%" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3,
                                                              100.0 * sum(v) / total))
