for cfg in "2 120000" "4 100000" "8 0" "12 0"; do
  set -- $cfg
  COMPEG_WPB=$1 COMPEG_LDS_PAD=$2 COMPEG_PIPELINE=split python bench.py --batch 128 --steps 4 --warmup 1 --cpu-seconds 0 --no-verify 2>gpurun_out/split.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('waves/CU $1', d['ms_per_frame']*1e3, d['roofline']['kernels_ms'])"
done
