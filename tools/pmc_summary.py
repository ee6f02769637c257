"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json (per-launch HBM bytes).

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced streaming reads, so the read side is reported both raw
and doubled (the traversal kernel's reads are 16-byte node / 4-byte ray gathers, an uncalibrated pattern: the truth lies
between the two).
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(dirname, counter, skip):
    per = defaultdict(list)
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v[skip:]) / max(len(v[skip:]), 1), len(v)) for k, v in per.items()}


def main(fetch_dir, write_dir, out, skip=265):
    fetch = load(fetch_dir, "FETCH_SIZE", skip)
    write = load(write_dir, "WRITE_SIZE", skip)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), steady-state dispatches only", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "gmupt" not in k:
            continue
        f = fetch.get(k, (0.0, 0))[0] * 1024.0
        w = write.get(k, (0.0, 0))[0] * 1024.0
        res["kernels"][k] = {"fetch_bytes_raw": int(f), "fetch_bytes_x2": int(2 * f), "write_bytes": int(w), "dispatches": fetch.get(k, (0, 0))[1]}
        for short in ("k_extend_d", "k_shadow_d", "k_cast_f", "k_cast_w", "k_cast_m", "k_cast_d"):
            if short in k:
                res[short + "_hbm_bytes_per_launch"] = int(2 * f + w)
                res[short + "_hbm_bytes_per_launch_raw"] = int(f + w)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 265)
