"""Per-kernel averages of every counter in a directory of rocprofv3 --pmc passes (steady-state dispatches only)."""
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 266
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gmupt" not in k: continue
        k = k.split("(")[0].replace("void gmupt::", "").replace("gmupt::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c][skip:]
        if v: print("   %-36s %16.1f" % (c, sum(v) / len(v)))
