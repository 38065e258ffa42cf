#!/bin/bash
# rocprofv3 passes over one extension-layout batch (tools/prof_layout.sh 2x2): kernel trace + two SQ counter passes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
S=${1:-2x2}
OUT=gpurun_out/prof_layout_$S
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --sampling $S --batch 64 --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-extra-configs --e2e-reps 0 --host-feed-ranks="
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1
python3 tools/pmc_summary.py $OUT decode_fused > $OUT/pmc_per_launch.txt 2>&1
cat $OUT/pmc_per_launch.txt
