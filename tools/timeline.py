"""Prints the GPU-side timeline (kernels and copies, microseconds from the first one) of the last
`--last N` operations found in a rocprofv3 --kernel-trace --memory-copy-trace CSV directory."""
import csv
import glob
import sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 14
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"] + " copy"))
ops.sort()
ops = ops[-last:]
t0 = ops[0][0]
prev_end = t0
for s, e, n in ops:
    print("%8.1f .. %8.1f  (%6.1f us, gap %5.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = max(prev_end, e)
