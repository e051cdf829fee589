"""per-kernel means of every counter in the rocprofv3 --pmc passes under a directory (tools/pmc_passes.sh)"""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "slavcheva"
for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for (k, c, gsz), v in sorted(acc.items()):
        print("%-62s grid %-8s %-28s n=%3d mean %.4g" % (k, gsz, c, len(v), sum(v) / len(v)))
