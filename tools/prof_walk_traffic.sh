#!/bin/bash
# HBM traffic of the walk + lane-per-MCU route's two kernels by the TCC counters (own passes):
#   tools/prof_walk_traffic.sh [frames] [ri]     (960x720 frames)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-256}; RI=${2:-60}
OUT=gpurun_out/prof_walk_${N}_$RI
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --width 960 --height 720 --ri $RI --batch $N --distinct 32 --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-extra-configs --e2e-reps 0 --host-feed-ranks="
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
grep -h '^{' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
python3 tools/pmc_summary.py $OUT 422 > $OUT/pmc_per_launch.txt 2>&1
cat $OUT/pmc_per_launch.txt
python3 -c "import json; d=json.load(open('$OUT/bench_under_profiler.json')); print('algorithmic bytes per launch', d['roofline']['algorithmic_bytes_per_launch'], 'kernel', d['roofline']['kernel'], d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
