"""Reduces the rocprofv3 outputs of tools/prof_bench.sh (gpurun_out/<tag>/) to the numbers bench.py and
DESIGN.md quote for the dominant kernel: average duration from --kernel-trace --stats, HBM-side traffic
from the FETCH_SIZE / WRITE_SIZE PMC passes (separate passes, as the microarchitecture guide prescribes),
cache hit rates from the TCP / TCC passes.

Units and gfx950 correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are reported in
KiB; on gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes, so the read side is doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        per launch (mean over the launches of the run)
Usage: python tools/pmc_traffic.py gpurun_out/r1f k_score profiles/r1_k_score_pmc.json [matches per launch, default 64]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
    res = {"kernel": kernel, "source": root, "matches_per_launch": float(sys.argv[4]) if len(sys.argv) > 4 else 64.0}
    for f in sorted(glob.glob(os.path.join(root, "pmc_*", "p_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, vals in agg.items():
            res[c] = sum(vals) / len(vals)
            res["launches_" + c] = len(vals)
    for r in csv.DictReader(open(os.path.join(root, "trace", "t_kernel_stats.csv"))):
        if kernel in r["Name"]:
            res["avg_ns"] = float(r["AverageNs"])
            res["calls"] = int(r["Calls"])
            res["kernel_name"] = r["Name"]
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        res["hbm_read_bytes_per_launch"] = 2.0 * res["FETCH_SIZE"] * 1024.0
        res["hbm_write_bytes_per_launch"] = res["WRITE_SIZE"] * 1024.0
        res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
        res["correction"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies 128-B requests at 64 B on gfx950)"
    if "TCC_HIT_sum" in res:
        res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in res:
        res["l1_hit_rate"] = 1.0 - res["TCP_TCC_READ_REQ_sum"] / res["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "SQ_INSTS_VMEM_RD" in res:
            res["tcp_accesses_per_vmem_instruction"] = res["TCP_TOTAL_CACHE_ACCESSES_sum"] / res["SQ_INSTS_VMEM_RD"]
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
