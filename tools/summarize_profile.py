#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_bench.sh (gpurun_out/<dir>) into the files committed under profiles/:
  <tag>_bench_kernel_stats.csv   the --kernel-trace --stats table, our kernels first
  <tag>_pmc_hbm_traffic.csv      per-kernel means of FETCH_SIZE / WRITE_SIZE (bench + calibration launches)
  traffic.json                   calibrated HBM bytes per launch of the fused kernel (read by bench.py)
usage: summarize_profile.py gpurun_out/<dir> <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
out_dir = os.path.join(ROOT, "profiles")
csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::)?(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def counter_means(sub):
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


rows = []
fused = {}
cal = {}
for run, subs in (("calibration", ("cal_fetch", "cal_write")), ("bench", ("pmc_fetch", "pmc_write"))):
    for sub in subs:
        for (kernel, counter), vals in sorted(counter_means(sub).items()):
            if "hier_update_kernel" in kernel and run == "calibration":
                cal[counter] = sum(vals) / len(vals)
            elif "slavcheva_iteration_kernel" in kernel and run == "bench":
                fused[counter] = (sum(vals) / len(vals), len(vals))
            else:
                continue
            rows.append((run, kernel, counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
with open(os.path.join(out_dir, tag + "_pmc_hbm_traffic.csv"), "w") as f:
    f.write("run,kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
    for r in rows:
        f.write('%s,"%s",%s,%d,%.1f,%.1f,%.1f\n' % r)

# calibration launches: hier_update_kernel<3> at 256^3 reads 393216 KiB and writes 196608 KiB (tools/pmc_calibrate.py)
fetch_corr = 393216.0 / cal["FETCH_SIZE"]
write_corr = 196608.0 / cal["WRITE_SIZE"]
hbm = (fused["FETCH_SIZE"][0] * fetch_corr + fused["WRITE_SIZE"][0] * write_corr) * 1024.0
json.dump(dict(workload="killing", size=256, hbm_bytes_per_launch=int(round(hbm)),
               fetch_size_KiB=fused["FETCH_SIZE"][0], fetch_correction=round(fetch_corr, 4),
               write_size_KiB=fused["WRITE_SIZE"][0], write_correction=round(write_corr, 4),
               dispatches=fused["FETCH_SIZE"][1],
               source="profiles/%s_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                      "calibrated on launches of known traffic)" % tag),
          open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
table = list(csv.DictReader(open(stats)))
with open(os.path.join(out_dir, tag + "_bench_kernel_stats.csv"), "w") as f:
    cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]
    f.write(",".join(cols) + "\n")
    for row in table:
        row = dict(row)
        row["Name"] = '"%s"' % short(row["Name"])
        f.write(",".join(str(row.get(c, "")) for c in cols) + "\n")
print(open(os.path.join(out_dir, "traffic.json")).read())
print(open(os.path.join(out_dir, tag + "_bench_kernel_stats.csv")).read()[:1500])
