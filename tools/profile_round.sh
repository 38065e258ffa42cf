#!/bin/bash
# Collects the rocprofv3 evidence for the bench command on the GPU box (run via gpurun):
#   1. --kernel-trace --stats  (per-kernel durations)
#   2. --pmc FETCH_SIZE        (own pass: 3 TCC slots)
#   3. --pmc WRITE_SIZE        (own pass)
#   4. --pmc SQ_* issue/wait counters (two passes)
# Counter passes never combine with trace domains other than --kernel-trace.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_round
rm -rf $OUT && mkdir -p $OUT
# (the stats pass: 24 timed launches of the headline kernel -- its average has to reproduce the bench line's
# kernel_ms -- and the sub-records, so that the single-frame and MJPEG-stream launches are in the trace too)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 24 --warmup 3 --cpu-seconds 0 --no-verify --e2e-reps 0 --no-sweep --host-feed-ranks "" > $OUT/stats.log 2>&1
B="python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-extra-configs --e2e-reps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1
grep -h '^{' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
ls -R $OUT | head -40
