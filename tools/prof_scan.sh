cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scan -- python3 bench.py --batch 128 --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --preprocess device-per-step > gpurun_out/prof_scan.log 2>&1
cat gpurun_out/prof_scan/*/*kernel_stats.csv | cut -c1-160
