#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes over `bench.py --workload sobolev` (tools/profile_bench.sh with BENCH_ARGS) ->
profiles/<tag>_sobolev_pmc_hbm_traffic.csv: per kernel of the SobolevFusion iteration the mean duration (kernel trace of
the stats pass), the calibrated HBM-side bytes per launch and the algorithmic bytes of SURVEY 8(d)'s 76 B split by pass.
usage: summarize_sobolev_pmc.py gpurun_out/<dir> <tag> [band voxels]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
band = int(sys.argv[3]) if len(sys.argv) > 3 else 1652481


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::)?(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def means(sub, column="Counter_Value"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row[column]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


cal = {}
for sub in ("cal_fetch", "cal_write"):
    m, _ = means(sub)
    for (kernel, counter), v in m.items():
        cal[(kernel.split("<")[0], counter)] = v
KNOWN = {("state_unpack_kernel", "FETCH_SIZE"): 262144.0, ("state_pack_kernel", "WRITE_SIZE"): 524288.0}
fetch_corr = KNOWN[("state_unpack_kernel", "FETCH_SIZE")] / cal[("state_unpack_kernel", "FETCH_SIZE")]
write_corr = KNOWN[("state_pack_kernel", "WRITE_SIZE")] / cal[("state_pack_kernel", "WRITE_SIZE")]
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
dur = {short(r["Name"]): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(stats))}
fetch, n_f = means("pmc_fetch")
write, _ = means("pmc_write")
# algorithmic bytes per band voxel and launch (fp32, D = 3; SURVEY 8d's S-sobolev 76 B = 20 + 12 | 12 + 12 | 12 + 12 ...
# on the float4 layouts the kernels move 16-byte gradients: both figures are listed)
ALG = {"sobolev_state_gradient_kernel": (20 + 12, "R state 16 + canonical 4, W raw gradient 12 (16 on the float4 layout)"),
       "sobolev_state_gradient_x_kernel": (20 + 12, "gradient AND x pass in one launch (round 4): R state 16 + canonical 4, W the x-filtered gradient 12 (16 on the float4 layout); the raw gradient stays in LDS"),
       "convolve_list4_kernel": (12 + 12, "R gradient 12 (+ mask 12 from L2), W 12 (16 + 16 on the float4 layout)"),
       "sobolev_state_update_kernel": (12 + 4 + 12 + 4, "R gradient 12 + live 4 (gathered), W warp 12 + live 4 (+ final gradient)"),
       "sobolev_state_box_kernel": (12 + 4 + 12 + 4 + 12 + 12, "y pass, z pass, update and re-warp in one launch, box by box (round 5): R the x-filtered gradient 12 (16) + live 4, W warp 12 + live 4; the y-filtered gradient (W 12 + R 12 of the budget) stays in LDS")}
out = os.path.join(ROOT, "profiles", tag + "_sobolev_pmc_hbm_traffic.csv")
with open(out, "w") as f:
    f.write("kernel,dispatches,mean_us,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_side_MB_calibrated,algorithmic_MB,what\n")
    total_us = total_hbm = 0.0
    for kernel in sorted(dur):
        base = kernel.split("<")[0]
        if base not in ALG:
            continue
        fk, wk = fetch.get((kernel, "FETCH_SIZE")), write.get((kernel, "WRITE_SIZE"))
        if fk is None or wk is None:
            continue
        hbm = (fk * fetch_corr + wk * write_corr) * 1024 / 1e6
        alg = ALG[base][0] * band / 1e6
        fused = any(k.startswith("sobolev_state_gradient_x_kernel") for k in dur)
        per_iter = (1 if fused else 2) if base == "convolve_list4_kernel" else 1
        total_us += dur[kernel][0] * per_iter
        total_hbm += hbm * per_iter
        f.write('"%s",%d,%.2f,%.1f,%.1f,%.1f,%.1f,"%s"\n' % (kernel, n_f[(kernel, "FETCH_SIZE")], dur[kernel][0], fk, wk, hbm, alg,
                                                         ALG[base][1]))
    f.write('"one iteration (gradient, filter passes, update)",,%.2f,,,%.1f,%.1f,"76 B x %d band voxels; fetch correction %.4f '
            'write correction %.4f (16-byte-per-lane calibration launches)"\n'
            % (total_us, total_hbm, 76 * band / 1e6, band, fetch_corr, write_corr))
print(open(out).read())
