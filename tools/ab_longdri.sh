#!/bin/bash
# batches of 256 960x720 frames by restart interval on the batch kernel
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, compeg_amd
gpu = compeg_amd.Gpu.open()
for ri in (1, 2, 4, 6, 8, 10, 16, 30, 60):
    r = bench.bench_config(compeg_amd, gpu, 960, 720, ri, 85, 256, 10, 3, 16, 32, "256 x 960x720 DRI=%d" % ri)
    print("DRI", ri, r["ms_per_step"], r["roofline"]["frac"], r["kernel"], r["verified_bit_exact_vs_oracle"], flush=True)
PY
