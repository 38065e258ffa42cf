cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python3 bench.py --batch 32 --steps 3 --warmup 1 --cpu-seconds 0 --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $B > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_pmc1 -- $B > gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_pmc2 -- $B > gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES --output-format csv -d gpurun_out/prof_pmc3 -- $B > gpurun_out/prof_pmc3.log 2>&1
ls -R gpurun_out/prof_stats | head -20
