#!/usr/bin/env python3
"""Summarise the rocprofv3 CSV output of scripts/profile_*.sh for one kernel.

    python scripts/summarize_prof.py gpurun_out/prof_<tag> [kernel-substring] [--json out.json]

Prints the kernel-trace statistics (calls / average / min duration) and the per-dispatch mean of every
PMC counter found in the pmc_* passes.  HBM bytes follow the gfx950 corrections: FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests of wide coalesced streams at 64 bytes, so the
read side is doubled.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = args[0]
    needle = args[1] if len(args) > 1 else "fbank512"
    out_json = None
    if "--json" in sys.argv:
        out_json = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != out_json]
    summary = {"kernel": needle}
    def newest(pattern):
        """One file per pass directory: the most recent one (a directory reused by several runs keeps the old CSVs)."""
        by_dir = {}
        for f in glob.glob(pattern):
            d = os.path.dirname(f)
            if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
                by_dir[d] = f
        return sorted(by_dir.values())

    for f in newest(os.path.join(root, "stats", "*", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            if needle in row["Name"]:
                summary["calls"] = int(row["Calls"])
                summary["avg_ns"] = float(row["AverageNs"])
                summary["min_ns"] = float(row["MinNs"])
                summary["max_ns"] = float(row["MaxNs"])
                summary["kernel_name"] = row["Name"]
    counters = defaultdict(list)
    meta = {}
    for f in newest(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"]:
                counters[row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
    summary["launch"] = meta
    summary["counters_mean_per_dispatch"] = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
    c = summary["counters_mean_per_dispatch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd = 2.0 * c["FETCH_SIZE"] * 1024.0
        wr = c["WRITE_SIZE"] * 1024.0
        summary["hbm_read_bytes_per_launch"] = rd
        summary["hbm_write_bytes_per_launch"] = wr
        summary["hbm_bytes_per_launch"] = rd + wr
    print(json.dumps(summary, indent=1))
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(summary, fh, indent=1)


if __name__ == "__main__":
    main()
