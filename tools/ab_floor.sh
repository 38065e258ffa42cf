#!/bin/bash
# VERDICT r03 item 5, by knock-outs: how much of the headline launch (256 x 4K DRI = 4) is vector work that could still be
# removed?  Laboratory builds (tools/build_variant.sh): base = as shipped; exp16 = no IDCT for the chroma data units (more
# than a per-wave sparse chroma transform can save); exp4 = no IDCT at all (-26 % of the vector instructions: more than the
# hand-written AC loop's -9 % and a sparse chroma transform together); exp9 / exp18 / exp17 = the same three with every
# global store knocked out.  Three rounds on one box, then one counter pass per build.
#   tools/build_variant.sh base ""; for v in 16 4 9 18 17; do tools/build_variant.sh exp$v "-DCG_EXP=$v"; done
#   gpurun --timeout 1200 -- bash tools/ab_floor.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_floor
rm -rf $OUT && mkdir -p $OUT
LIBS="base exp16 exp4 exp9 exp18 exp17"
ARGS="--steps ${STEPS:-12} --warmup 3 --cpu-seconds 0 --no-verify --e2e-reps 0 --no-sweep --no-extra-configs --host-feed-ranks ''"
for round in 1 2 3; do
  for l in $LIBS; do
    COMPEG_LIB=$PWD/gpurun_ab/lib_$l.so timeout -k 10 300 python3 bench.py --steps ${STEPS:-12} --warmup 3 --cpu-seconds 0 --no-verify --e2e-reps 0 --no-sweep --no-extra-configs --host-feed-ranks "" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('round $round %-6s kernel_ms %.4f  ms_per_step %.4f' % ('$l', d['roofline']['kernel_ms'], d['ms_per_step']), flush=True)" | tee -a $OUT/rounds.txt || exit 1
  done
done
for l in $LIBS; do
  COMPEG_LIB=$PWD/gpurun_ab/lib_$l.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$l -- python3 bench.py --steps 3 --warmup 1 --prime-seconds 0 --cpu-seconds 0 --no-verify --e2e-reps 0 --no-sweep --no-extra-configs --host-feed-ranks "" > $OUT/pmc_$l.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for l in "base exp16 exp4 exp9 exp18 exp17".split():
    for p in glob.glob(f'gpurun_out/ab_floor/pmc_{l}/*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if 'decode_fused_422_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) >= 256 * 64:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        # the headline launches are the largest ones: the last three of each counter
        print(l, {c: round(sum(sorted(v)[-3:]) / 3) for c, v in acc.items()})
PY
