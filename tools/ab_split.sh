#!/bin/bash
# like ab_bench.sh, for the two-kernel pipeline: prints per-kernel times
ARGS="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    COMPEG_PIPELINE=split COMPEG_LIB="$PWD/$lib" python bench.py $ARGS --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-28s us/frame %.3f  %s' % ('$lib', d['ms_per_frame']*1e3, d['roofline']['kernels_ms']))"
  done
done
