#!/bin/bash
# the 8K DRI=1 batch (BASELINE configs[4]): rows leaving wave-wide against the quad exchange (laboratory library)
cd "$GRAFT_REPO_ROOT"
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so
for rep in 1 2; do for wide in 1 0; do
COMPEG_WIDE=$wide python3 - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, compeg_amd
gpu = compeg_amd.Gpu.open()
r = bench.bench_config(compeg_amd, gpu, 7680, 4320, 1, 85, 8, 30, 5, 16, 4, "8 x 8K DRI=1")
print("wide", os.environ["COMPEG_WIDE"], json.dumps({k: r[k] for k in ("ms_per_step", "roofline", "verified_bit_exact_vs_oracle")}))
r = bench.bench_config(compeg_amd, gpu, 1920, 1080, 1, 85, 256, 30, 5, 16, 16, "256 x 1080p DRI=1")
print("wide", os.environ["COMPEG_WIDE"], json.dumps({k: r[k] for k in ("ms_per_step", "roofline", "verified_bit_exact_vs_oracle")}))
PY
done; done
