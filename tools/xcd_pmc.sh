#!/bin/bash
# L2 counters of the three legs of tools/xcd_experiment.py (the last 9 dispatches of k_cast_w of each run are the timed ones)
export TMPDIR=/tmp
OUT=gpurun_out/xcd_pmc; rm -rf $OUT; mkdir -p $OUT
for mode in base binned xcd; do
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ --output-format csv -d $OUT/$mode -- python3 tools/xcd_experiment.py --pmc-mode $mode > $OUT/$mode.log 2>&1 || { echo "$mode failed"; tail -3 $OUT/$mode.log; exit 1; }
  python3 - $OUT/$mode $mode <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_cast_w" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
out = {}
for c, v in acc.items():
    v.sort(); last = [x for _, x in v[-9:]]; out[c] = sum(last) / len(last)
print("%-7s TCC_REQ %.1f M  TCC_HIT %.1f M  hit rate %.3f  memory-side read requests %.1f M" % (sys.argv[2], out["TCC_REQ"] / 1e6, out["TCC_HIT"] / 1e6, out["TCC_HIT"] / out["TCC_REQ"], out["TCC_EA0_RDREQ"] / 1e6))
PY
  rm -rf $OUT/$mode
done
