#!/bin/bash
# A/B of library builds over several batch sizes (run via gpurun): tools/ab_batches.sh "4 8 16 32 64" libA.so libB.so
SIZES="$1"; shift
for b in $SIZES; do
  for lib in "$@"; do
    COMPEG_LIB="$PWD/$lib" python bench.py --batch $b --steps 6 --warmup 2 --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('batch %-4s %-26s us/frame %.2f  ms/step %.3f' % ('$b', '$lib', d['ms_per_frame']*1e3, d['ms_per_step']))"
  done
done
