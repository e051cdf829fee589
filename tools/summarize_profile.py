#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_bench.sh (gpurun_out/<dir>) into the files committed under profiles/:
  <tag>_bench_kernel_stats.csv   the --kernel-trace --stats table, our kernels first
  <tag>_pmc_hbm_traffic.csv      per-kernel means of FETCH_SIZE / WRITE_SIZE (bench + calibration launches)
  traffic.json                   calibrated HBM bytes per launch of the fused kernel (read by bench.py)
usage: summarize_profile.py gpurun_out/<dir> <tag> [size [traffic file name]]
  size (default 256): the bench's --size of the profiled run; a size other than 256 writes traffic_<size>.json
  a directory without PMC passes (PASSES=stats) only yields the kernel-stats table"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
traffic_name = sys.argv[4] if len(sys.argv) > 4 else ("traffic.json" if size == 256 else "traffic_%d.json" % size)
out_dir = os.path.join(ROOT, "profiles")


KERNEL_AVERAGES = {}  # walk -> (rocprofv3's average duration of the fused kernel in us, dispatches)


def write_stats():
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    table = list(csv.DictReader(open(stats)))
    for row in table:
        name = short(row["Name"])
        if name.startswith(("slavcheva_state_kernel", "slavcheva_state_box_kernel")):
            # the band walk of the step: the list kernel, or the box kernel where the engine picks it (512^3)
            walk = "dense" if name.startswith("slavcheva_state_kernel") and name.rstrip(">").endswith(" 0") else "list"
            if walk not in KERNEL_AVERAGES or int(row["Calls"]) > KERNEL_AVERAGES[walk][1]:
                KERNEL_AVERAGES[walk] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    with open(os.path.join(out_dir, tag + "_bench_kernel_stats.csv"), "w") as f:
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]
        f.write(",".join(cols) + "\n")
        for row in table:
            row = dict(row)
            row["Name"] = '"%s"' % short(row["Name"])
            f.write(",".join(str(row.get(c, "")) for c in cols) + "\n")
    print(open(os.path.join(out_dir, tag + "_bench_kernel_stats.csv")).read()[:1500])
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


if not os.path.isdir(os.path.join(src, "pmc_fetch")):
    write_stats()
    sys.exit(0)

rows = []
fused = {"list": {}, "dense": {}}
cal = {}
for run, subs in (("calibration", ("cal_fetch", "cal_write")), ("bench", ("pmc_fetch", "pmc_write"))):
    for sub in subs:
        for (kernel, counter), vals in sorted(counter_means(sub).items()):
            mean = sum(vals) / len(vals)
            if run == "calibration" and kernel.startswith(("hier_update_kernel", "state_unpack_kernel",
                                                           "state_pack_kernel")):
                cal[(kernel.split("<")[0], counter)] = mean
            elif run == "bench" and ("slavcheva_state_kernel" in kernel or "slavcheva_state_box_kernel" in kernel):
                # last template argument: 0 dense walk, 1 list, 2 interior list (the bench's band list); the box kernel
                # stands for the band walk where the engine picks it
                walk = "dense" if "slavcheva_state_kernel" in kernel and kernel.rstrip(">").endswith(" 0") else "list"
                fused[walk][counter] = (mean, len(vals))
            else:
                continue
            rows.append((run, kernel, counter, len(vals), mean, min(vals), max(vals)))
with open(os.path.join(out_dir, tag + "_pmc_hbm_traffic.csv"), "w") as f:
    f.write("run,kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
    for r in rows:
        f.write('%s,"%s",%s,%d,%.1f,%.1f,%.1f\n' % r)

# calibration launches (tools/pmc_calibrate.py), KiB actually moved per launch at 256^3
KNOWN = {("hier_update_kernel", "FETCH_SIZE"): 393216.0, ("hier_update_kernel", "WRITE_SIZE"): 196608.0,
         ("state_unpack_kernel", "FETCH_SIZE"): 262144.0, ("state_pack_kernel", "WRITE_SIZE"): 524288.0}
corr = {k: KNOWN[k] / v for k, v in cal.items() if k in KNOWN}
# the state kernel moves 16 bytes per lane: corrections of the 16-byte calibration launches
fetch_corr = corr[("state_unpack_kernel", "FETCH_SIZE")]
write_corr = corr[("state_pack_kernel", "WRITE_SIZE")]
sys.path.insert(0, ROOT)
import levelsetfusion_python_amd as _pkg  # noqa: E402  the build the counters were collected on (same sources)

result = dict(workload="killing", size=size, tag=tag, build_id=_pkg._lib.lib.lsf_build_id().decode(), fetch_correction=round(fetch_corr, 4), write_correction=round(write_corr, 4),
              corrections={"%s %s" % k: round(v, 4) for k, v in corr.items()},
              source="profiles/%s_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                     "calibrated on launches of known traffic with the same bytes per lane)" % tag)
for walk, key in (("list", "hbm_bytes_per_launch"), ("dense", "dense_hbm_bytes_per_launch")):
    c = fused[walk]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        result[key] = int(round((c["FETCH_SIZE"][0] * fetch_corr + c["WRITE_SIZE"][0] * write_corr) * 1024.0))
        result[walk + "_fetch_size_KiB"] = c["FETCH_SIZE"][0]
        result[walk + "_write_size_KiB"] = c["WRITE_SIZE"][0]
        result[walk + "_dispatches"] = c["FETCH_SIZE"][1]
write_stats()
# rocprofv3's own per-dispatch average of the fused kernel (the --kernel-trace --stats pass of the same command): what the
# judge recomputes the roofline fraction from; bench.py prints it next to its HIP-event figure while the build matches
for walk, key in (("list", "list_kernel_avg_us"), ("dense", "dense_kernel_avg_us")):
    if walk in KERNEL_AVERAGES:
        result[key], result[key.replace("avg_us", "dispatches")] = KERNEL_AVERAGES[walk]
result["kernel_stats"] = "profiles/%s_bench_kernel_stats.csv" % tag
json.dump(result, open(os.path.join(out_dir, traffic_name), "w"), indent=1)
print(open(os.path.join(out_dir, traffic_name)).read())
