#!/bin/bash
# Calibrates rocprofv3 FETCH_SIZE / TCC_EA0_RDREQ for RANDOM record gathers (MI355X_MICROARCH.md "HBM": uncalibrated outside wide streaming
# reads).  Known byte counts: tools/micro/gather64 on a 1 GB table (every record a fresh line; 3 % L2 hits by size) -- shape A fetches
# 64-byte aligned records, shape D 48-byte records at a 48-byte stride, which touch 1.5 sixty-four-byte blocks but only 1.25 128-byte lines
# per record on average: requests per record tell the granule of a memory-side read.  hbm_copy is the streaming reference.
# Usage (GPU box): tools/micro/calibrate_fetch.sh gpurun_out/<dir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/calib}; mkdir -p $OUT
for set in "FETCH_SIZE" "TCC_EA0_RDREQ TCC_MISS TCC_REQ TCC_HIT" "WRITE_SIZE" "TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_DRAM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/g_$tag -- ./tools/micro/gather64 16000000 > $OUT/g_$tag.log 2>&1 || { echo "gather pass $tag failed"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/c_$tag -- ./tools/micro/hbm_copy > $OUT/c_$tag.log 2>&1 || { echo "copy pass $tag failed"; exit 1; }
done
python3 - $OUT <<'PY' > $OUT/calibration.txt
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
recs = 2048 * 256 * 64   # lane-records per gather64 launch
print("gather64, 1 GB table: %d records per launch; hbm_copy: 2 GiB read + 2 GiB written (copy), 4 GiB read + 2 GiB written (triad)" % recs)
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    line = "%-28s" % k[:28]
    if "FETCH_SIZE" in c: line += "  FETCH_SIZE %.3f GB" % (c["FETCH_SIZE"] * 1024 / 1e9)
    if "WRITE_SIZE" in c: line += "  WRITE_SIZE %.3f GB" % (c["WRITE_SIZE"] * 1024 / 1e9)
    if "TCC_EA0_RDREQ_128B" in c: line += "  EA reads by size: 32B %.1f M  64B %.1f M  128B %.1f M = %.3f GB; to DRAM %.1f M" % (c["TCC_EA0_RDREQ_32B"] / 1e6, c["TCC_EA0_RDREQ_64B"] / 1e6, c["TCC_EA0_RDREQ_128B"] / 1e6, (32 * c["TCC_EA0_RDREQ_32B"] + 64 * c["TCC_EA0_RDREQ_64B"] + 128 * c["TCC_EA0_RDREQ_128B"]) / 1e9, c["TCC_EA0_RDREQ_DRAM"] / 1e6)
    if "TCC_EA0_RDREQ" in c: line += "  EA_RDREQ %.1f M (%.3f per record)  TCC_REQ %.1f M  MISS %.1f M  HIT %.1f M" % (c["TCC_EA0_RDREQ"] / 1e6, c["TCC_EA0_RDREQ"] / recs, c["TCC_REQ"] / 1e6, c["TCC_MISS"] / 1e6, c["TCC_HIT"] / 1e6)
    print(line)
PY
rm -rf $OUT/g_* $OUT/c_*
cat $OUT/calibration.txt
