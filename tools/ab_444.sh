#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so
B="python3 bench.py --batch 64 --steps 10 --warmup 3 --cpu-seconds 0 --no-verify --no-extra-configs --e2e-reps 0 --host-feed-ranks="
for s in ${SAMPLINGS:-1x1 1x2}; do
for w in ${WPBS:-0 2 3 4 5}; do
  if [ $w = 0 ]; then unset COMPEG_WPB; else export COMPEG_WPB=$w; fi
  echo "sampling $s wpb=$w: $(COMPEG_VERBOSE=1 $B --sampling $s 2>&1 | grep -E 'plan: images=64|^\{' | sed -E 's/.*(plan: [^\n]*)/\1/; s/.*"ms_per_step": ([0-9.]*).*/ms_per_step \1/' | sort -u | tr '\n' ' ')"
done; done
