#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_round.sh.

    python tools/pmc_summary.py gpurun_out/prof_round [kernel-substring]

Prints, per (kernel, grid size): launches, average duration from the kernel
trace, and the per-launch mean of every PMC counter found in the counter
passes.  TCC byte counters are converted as MI355X_MICROARCH.md prescribes
(FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE x2 for streaming reads).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    trace = newest(os.path.join(root, "stats", "**", "*_kernel_trace.csv"))
    dur = defaultdict(list)
    if trace:
        for row in csv.DictReader(open(trace)):
            grid = int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"])
            dur[(row["Kernel_Name"], grid)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    counters = defaultdict(lambda: defaultdict(list))
    meta = {}
    for d in sorted(os.listdir(root)):
        f = newest(os.path.join(root, d, "**", "*_counter_collection.csv"))
        if not f:
            continue
        per_dispatch = defaultdict(float)
        info = {}
        for row in csv.DictReader(open(f)):
            key = (row["Kernel_Name"], int(row["Grid_Size"]), row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            info[(row["Kernel_Name"], int(row["Grid_Size"]))] = (
                row["Workgroup_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row["Scratch_Size"])
        for (k, g, _, c), v in per_dispatch.items():
            counters[(k, g)][c].append(v)
        meta.update(info)
    keys = sorted(set(dur) | set(counters))
    for k in keys:
        name, grid = k
        if want not in name:
            continue
        print(f"## {name.split('(')[0]}  grid={grid}")
        if k in meta:
            wg, lds, vgpr, scratch = meta[k]
            print(f"   workgroup {wg}, LDS/block {lds}, VGPR {vgpr}, scratch {scratch}")
        if k in dur:
            v = dur[k]
            print(f"   launches {len(v)}, avg {sum(v) / len(v) / 1e3:.1f} us, min {min(v) / 1e3:.1f} us")
        for c, v in sorted(counters.get(k, {}).items()):
            mean = sum(v) / len(v)
            extra = ""
            if c == "WRITE_SIZE":
                extra = f"  = {mean * 1024 / 1e9:.4f} GB"
            if c == "FETCH_SIZE":
                extra = f"  x2 (gfx950 streaming-read correction) = {mean * 2 * 1024 / 1e9:.4f} GB"
            print(f"   {c:24s} {mean:16.0f}{extra}")


if __name__ == "__main__":
    main()
