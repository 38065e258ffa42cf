#!/bin/bash
# batches of 256 960x720 frames by restart interval: whole-interval windows (COMPEG_STREAM=0) against the streamed
# window (1) and what the library picks (laboratory library)
cd "$GRAFT_REPO_ROOT"
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so
for mode in ${MODES:-auto 0 1}; do
if [ $mode = auto ]; then unset COMPEG_STREAM; else export COMPEG_STREAM=$mode; fi
timeout -k 10 600 python3 - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, compeg_amd
gpu = compeg_amd.Gpu.open()
for ri in [int(x) for x in os.environ.get("DRIS", "1 4 6 8 10 16 30 60").split()]:
    r = bench.bench_config(compeg_amd, gpu, 960, 720, ri, 85, 256, 10, 3, 16, 32, "256 x 960x720 DRI=%d" % ri)
    print("stream", os.environ.get("COMPEG_STREAM", "auto"), "rows", os.environ.get("COMPEG_STREAM_ROWS", "-"), "DRI", ri, r["ms_per_step"], r["roofline"]["frac"], r["kernel"], r["verified_bit_exact_vs_oracle"], flush=True)
PY
done
