"""tools/floor_table.sh's passes -> one table: per variant of the fused iteration kernel, the per-launch means of the
counters and the ratios that say what binds it (VALU issue, vector-memory issue, the L1's tag path, waiting)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
rows = []
for d in sorted(glob.glob(os.path.join(root, "n*_e*"))):
    name = os.path.basename(d)
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "slavcheva_state" in r["Kernel_Name"] and "prepare" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    avg_us = None
    for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "slavcheva_state" in r["Name"] and ("box_kernel" in r["Name"] or "slavcheva_state_kernel" in r["Name"]):
                avg_us = float(r["AverageNs"]) / 1e3
    events = open(os.path.join(d, "events.txt")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(d, "events.txt")) else ""
    rows.append((name, c, avg_us, events))
print("What binds the fused KillingFusion iteration kernel (MI355X, sphere pair; per-launch means of rocprofv3 --pmc passes,")
print("one counter group per pass; tools/floor_table.sh).  n<N>_<walk>_e<energy sums on / off>.")
print()
for name, c, avg_us, events in rows:
    print("== %s" % name)
    print("   %s" % events)
    if avg_us:
        print("   rocprofv3 --stats average per dispatch: %.2f us" % avg_us)
    for k in sorted(c):
        print("   %-34s %14.6g" % (k, c[k]))
    g = c.get
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        # SQ_BUSY_CYCLES counts per SE (x 32 on this part in the summed output); ratios of per-wave counters are what is robust
        wc = g("SQ_WAVE_CYCLES", 0.0)
        print("   -- of a wave's resident cycles (SQ_WAVE_CYCLES): VALU issuing %.1f %%, vector memory issuing %.1f %%, LDS %.1f %%, "
              "scalar %.1f %%, waiting for an instruction's operands / issue (SQ_WAIT_INST_ANY) %.1f %%, any wait %.1f %%"
              % (100 * g("SQ_ACTIVE_INST_VALU", 0) / wc if wc else 0, 100 * g("SQ_ACTIVE_INST_VMEM", 0) / wc if wc else 0,
                 100 * g("SQ_ACTIVE_INST_LDS", 0) / wc if wc else 0, 100 * g("SQ_ACTIVE_INST_SCA", 0) / wc if wc else 0,
                 100 * g("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, 100 * g("SQ_WAIT_ANY", 0) / wc if wc else 0))
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print("   -- per wave: %.0f VALU, %.0f SALU, %.0f vector loads, %.0f vector stores, %.0f LDS instructions"
              % (g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU", 0) / g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD", 0) / g("SQ_WAVES"),
                 g("SQ_INSTS_VMEM_WR", 0) / g("SQ_WAVES"), g("SQ_INSTS_LDS", 0) / g("SQ_WAVES")))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("GRBM_GUI_ACTIVE"):
        print("   -- vector L1: %.2f tag accesses per CU and clock (256 CUs, GRBM_GUI_ACTIVE clocks), %.1f %% of its cycles stalled "
              "on pending data" % (g("TCP_TOTAL_CACHE_ACCESSES_sum") / 256.0 / g("GRBM_GUI_ACTIVE"),
                                   100 * g("TCP_PENDING_STALL_CYCLES_sum", 0) / max(g("TCP_GATE_EN1_sum", 0), 1)))
    print()
