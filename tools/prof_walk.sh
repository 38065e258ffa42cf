#!/bin/bash
# Issue / stall counters of the walk + lane-per-MCU route's two kernels on one batch (PMC passes, kernel trace only).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_walk
rm -rf $OUT && mkdir -p $OUT
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so COMPEG_WALK=${WALK:-1} PROBE_REPS=4 PROBE_CHECK=0
B="python3 tools/walk_probe.py ${CONFIGS:-960x720:60:256}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/c -- $B > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/d -- $B > $OUT/d.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/prof_walk/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0]
        if 'walk_mcus' in k or 'decode_fused' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(p.split('/')[2], k[:40], {c: round(sum(v[-2:]) / len(v[-2:])) for c, v in cs.items()})
PY
