#!/usr/bin/env python3
"""The figures the round's documents quote, from the outputs of tools/profile_round.sh and a bench line:

    python tools/round_figures.py gpurun_out/prof_round gpurun_out/bench_r03_final.json

Per (decode kernel, grid): launches, average / min / max duration in the newest kernel trace, first and last launches;
the headline kernel's last `steps` launches (the timed ones: clock priming and warm-up run in front of them) against the
bench-under-profiler line's kernel_ms; the bench line's headline and sub-record figures.
"""
import csv
import glob
import json
import os
import sys


def last_json_line(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    root, bench_path = sys.argv[1], sys.argv[2]
    trace = max(glob.glob(os.path.join(root, "stats", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    by = {}
    for r in csv.DictReader(open(trace)):
        key = (r["Kernel_Name"].split("(")[0].replace("compeg::", ""), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]),
               int(r["Workgroup_Size_X"]))
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
    print("## trace:", trace)
    for k, v in sorted(by.items()):
        if k[0].startswith("decode") or k[0].startswith("walk_"):
            print("%-34s grid %8d x %4d launches %4d avg %8.1f min %8.1f max %8.1f us  first %s last %s" % (
                k[0], k[1] // k[2], k[2], len(v), sum(v) / len(v), min(v), max(v), [round(x) for x in v[:4]], [round(x) for x in v[-3:]]))
    under = last_json_line(os.path.join(root, "bench_under_profiler.json"))
    steps = under["steps"]
    head = [v for k, v in by.items() if k[0] == "decode_fused_422_kernel" and k[1] // k[2] == 256 and k[2] == 768]
    head = max(head, key=lambda v: sum(v) / len(v)) if head else []
    if head:
        timed = head[-steps:]
        print("## headline kernel: %d launches, the last %d (timed) avg %.1f min %.1f max %.1f us; in front of them: %s" % (
            len(head), steps, sum(timed) / len(timed), min(timed), max(timed), [round(x) for x in head[:-steps]]))
        print("## bench under the profiler: kernel_ms %.4f frac %.4f value %.1f -> trace / bench = %.4f" % (
            under["roofline"]["kernel_ms"], under["roofline"]["frac"], under["value"],
            sum(timed) / len(timed) / 1000 / under["roofline"]["kernel_ms"]))
    d = last_json_line(bench_path)
    print("## bench line: value %.1f ms_per_step %.4f frac %.4f kernel_ms %.4f steps %d warmup %d prime %s" % (
        d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["steps"], d["warmup"], d.get("clock_prime")))
    sf = d.get("single_frame") or {}
    print("single_frame:", {k: v for k, v in sf.items() if not isinstance(v, dict)})

    def show(k, v, ind="  "):
        if isinstance(v, dict) and "roofline" in v:
            print(ind, k, "frac", v["roofline"].get("frac"), "kernel_ms", v["roofline"].get("kernel_ms"), "warm", v.get("warmup_decodes"),
                  v.get("kernel"), "value", v.get("value"))
        elif isinstance(v, dict):
            print(ind, k, {a: b for a, b in v.items() if not isinstance(b, (dict, list))})
            for a, b in v.items():
                if isinstance(b, dict):
                    show(a, b, ind + "  ")
        else:
            print(ind, k, v)
    for k, v in (d.get("other_configs") or {}).items():
        show(k, v)
    e2e = d.get("end_to_end") or {}
    for k, v in e2e.items():
        if isinstance(v, dict) and ("value" in v or "mpix_s" in v):
            print("  end_to_end", k, {a: b for a, b in v.items() if not isinstance(b, (dict, list))})


if __name__ == "__main__":
    main()
