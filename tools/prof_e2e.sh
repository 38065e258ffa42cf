cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_e2e
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_e2e -- python3 tools/e2e_probe.py > gpurun_out/prof_e2e.log 2>&1
cat gpurun_out/prof_e2e/*/*kernel_stats.csv | cut -c1-200
python3 tools/timeline.py gpurun_out/prof_e2e 12
