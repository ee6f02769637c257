"""Derives the roofline figures of a profile directory (profiles/<name>/) from the files in it and writes roofline.json there.

Inputs: run.json (plain run: HIP-event launch time, counted walk statistics), kernel_stats.csv (rocprofv3 --kernel-trace --stats: average
launch duration), pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes), ea_read_sizes.txt (TCC_EA0_RDREQ_{32B,64B,128B} pass).
Memory-side read bytes are taken from the request-size counters (every read request of these kernels is a 128-byte line:
profiles/r02_micro/fetch_size_calibration.txt); FETCH_SIZE tallies such a request at 64 bytes, so it equals HALF of them.
"""
import csv, json, re, sys
d = sys.argv[1].rstrip("/")
run = json.load(open(d + "/run.json"))
cast = run.get("cast") or run.get("roofline")
ms_event = cast["avg_launch_ms"]
# the kernel that was profiled: the one the plain run names (k_cast_w = the wide ray cast, the default; k_cast_f = the binary fused one)
KERNEL = cast.get("kernel") or ("k_cast_w" if (run.get("kernel_flags") or {}).get("k_cast_w") else "k_cast_f")
ms_trace = None
for row in csv.DictReader(open(d + "/kernel_stats.csv")):
    if KERNEL in row["Name"]:
        ms_trace = float(row["AverageNs"]) / 1e6; calls = int(row["Calls"])
traffic = json.load(open(d + "/pmc_traffic.json"))
k = [v for n, v in traffic["kernels"].items() if KERNEL in n][0]
sizes = {}
cur = None
for line in open(d + "/ea_read_sizes.txt"):
    if not line.startswith(" "): cur = line.strip(); continue
    if cur and cur.startswith(KERNEL):
        name, val = line.split(); sizes[name] = float(val)
read_bytes = 32 * sizes["TCC_EA0_RDREQ_32B"] + 64 * sizes["TCC_EA0_RDREQ_64B"] + 128 * sizes["TCC_EA0_RDREQ_128B"]
write_bytes = k["write_bytes"]
out = {"kernel": KERNEL, "avg_launch_ms_hip_events": ms_event, "avg_launch_ms_rocprof_trace": round(ms_trace, 4), "trace_calls": calls,
       "memory_side": {"read_requests_128B": int(sizes["TCC_EA0_RDREQ_128B"]), "read_bytes": int(read_bytes), "FETCH_SIZE_bytes_raw": k["fetch_bytes_raw"],
                       "write_bytes": write_bytes, "bytes_per_launch": int(read_bytes + write_bytes),
                       "gbs": round((read_bytes + write_bytes) / (ms_trace * 1e-3) / 1e9, 1), "hbm_peak_gbs": 8000.0,
                       "frac_of_hbm_peak": round((read_bytes + write_bytes) / (ms_trace * 1e-3) / 1e9 / 8000.0, 4),
                       "note": "Infinity-Cache hits are inside these counters (they count the L2's memory-side requests)"},
       "records": {"per_launch": cast.get("global_records_per_launch") or cast.get("records_per_launch"),
                   "grecords_per_s": cast.get("grecords_per_s") or cast.get("achieved")}}
# vector-memory request rate: L1 (TCP) accesses per clock and CU, from the per-kernel averages of the counter passes
pmc, cur = {}, None
try:
    for line in open(d + "/pmc_counters.txt"):
        if not line.startswith(" "): cur = line.strip(); continue
        if cur and cur.startswith(KERNEL):
            name, val = line.split(); pmc[name] = float(val)
except FileNotFoundError:
    pass
if "TCP_TOTAL_CACHE_ACCESSES" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    CUS, XCDS = 256, 8
    clocks = pmc["GRBM_GUI_ACTIVE"] / XCDS            # the counter is summed over the 8 XCDs
    out["vmem_request_rate"] = {"l1_accesses_per_launch": int(pmc["TCP_TOTAL_CACHE_ACCESSES"]), "gpu_clocks_per_launch": int(clocks),
                                "l1_accesses_per_clk_cu": round(pmc["TCP_TOTAL_CACHE_ACCESSES"] / clocks / CUS, 3),
                                "l1_to_l2_read_requests": int(pmc.get("TCP_TCC_READ_REQ", 0)),
                                "l2_hit_rate": round(pmc["TCC_HIT"] / pmc["TCC_REQ"], 3) if pmc.get("TCC_REQ") else None,
                                "ceiling_note": "tools/micro/gather64 shape A: 1.05 per clock and CU with every record an L2 hit, 0.75 for a uniform 18.6 MB table (profiles/r02_micro/)"}
json.dump(out, open(d + "/roofline.json", "w"), indent=1)
print(json.dumps(out, indent=1))
