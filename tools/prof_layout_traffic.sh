#!/bin/bash
# HBM traffic of the fused extension-layout kernels by the TCC counters (own passes, as MI355X_MICROARCH.md prescribes):
#   tools/prof_layout_traffic.sh 1x1   (4:4:4)      tools/prof_layout_traffic.sh 2x2   (4:2:0)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
S=${1:-2x2}
OUT=gpurun_out/prof_layout_traffic_$S
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --sampling $S --batch 64 --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-extra-configs --e2e-reps 0 --host-feed-ranks="
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
grep -h '^{' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
python3 tools/pmc_summary.py $OUT decode_fused > $OUT/pmc_per_launch.txt 2>&1
cat $OUT/pmc_per_launch.txt
