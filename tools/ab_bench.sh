#!/bin/bash
# A/B timing of library builds on one GPU box (run via gpurun):
#   tools/ab_bench.sh "<bench args>" libA.so libB.so [...]
# Runs bench.py with each library in turn, three rounds, so that clock drift
# between runs shows up as spread inside a variant rather than as a difference.
ARGS="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do
    COMPEG_LIB="$PWD/$lib" python bench.py $ARGS --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-28s us/frame %.3f  kernel_ms %.4f  single %.1f us' % ('$lib', d['ms_per_frame']*1e3, d['roofline']['kernel_ms'], d['single_frame']['kernel_ms']*1e3))"
  done
done
