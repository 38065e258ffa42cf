#!/bin/bash
# Timeline of single-frame decodes: every HIP call (host side), kernel and copy (device side) of 20 blocking decodes and of 20
# back-to-back one-image batch decodes.  No counters (gpurun refuses --pmc with the trace domains).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_single
rm -rf $OUT && mkdir -p $OUT
python3 tools/single_frame_probe.py $1 > $OUT/untraced.txt 2>&1
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 tools/single_frame_probe.py $1 > $OUT/traced.txt 2>&1
python3 - <<'PY'
import csv, glob
ops = []
for f in glob.glob('gpurun_out/prof_single/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'GPU  ' + r['Kernel_Name'].split('(')[0][-44:]))
for f in glob.glob('gpurun_out/prof_single/t/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'GPU  copy ' + r['Direction']))
for f in glob.glob('gpurun_out/prof_single/t/**/*hip_api_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'host ' + r['Function']))
ops.sort()
# the last 20 decode kernels are the batch's, the 20 in front of them (skipping the batch's warm-up) the decoder's
kern = [i for i, o in enumerate(ops) if o[2].startswith('GPU') and 'decode_' in o[2]]
def show(title, first, last):
    print('##', title)
    t0 = ops[first][0]
    for s, e, n in ops[first:last + 1]:
        print('%9.1f .. %9.1f  (%6.1f us)  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
if len(kern) >= 100:
    # decoder: decodes 30 warm + 20 timed; batch: 30 + 20 (+ 30 + 20 without timing events)
    has_untimed = len(kern) >= 150
    b_end = kern[99]
    show('three blocking decodes (Decoder), everything between the kernels of decodes 46 and 49', kern[45] + 1, kern[48])
    show('three back-to-back decodes of a one-image Batch', kern[95] + 1, kern[98])
    if has_untimed:
        show('... without timing events', kern[145] + 1, kern[148])
    gaps = [(ops[kern[i + 1]][0] - ops[kern[i]][1]) / 1e3 for i in range(80, 99)]
    durs = [(ops[kern[i]][1] - ops[kern[i]][0]) / 1e3 for i in range(80, 100)]
    print('## batch: kernel %.1f us, gap between two %.1f us (medians)' % (sorted(durs)[10], sorted(gaps)[9]))
    if has_untimed:
        gaps = [(ops[kern[i + 1]][0] - ops[kern[i]][1]) / 1e3 for i in range(130, 149)]
        print('## batch without timing events: gap between two %.1f us (median)' % sorted(gaps)[9])
PY
cat $OUT/untraced.txt
