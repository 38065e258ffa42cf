#!/bin/bash
# the walk + lane-per-MCU route (COMPEG_WALK=1) against what the library otherwise picks (0), laboratory library
cd "$GRAFT_REPO_ROOT"
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so
for mode in ${MODES:-1 0}; do
  export COMPEG_WALK=$mode
  echo "== COMPEG_WALK=$mode"
  timeout -k 10 ${TMO:-500} python3 tools/walk_probe.py $CONFIGS || exit 1
done
