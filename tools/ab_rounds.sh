#!/bin/bash
# blocking 4K decode against the host scan's rounds / the calling thread's share (laboratory library)
cd "$GRAFT_REPO_ROOT"
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so
for rounds in 2 3 4 6; do for own in 0 1 2 4; do
  echo "rounds=$rounds own=$own: $(COMPEG_SCAN_ROUNDS=$rounds COMPEG_SCAN_OWN=$own python3 tools/e2e_probe.py host 2>&1 | grep -E 'decode_blocking|start_decode' | tr '\n' ' ')"
done; done
echo "threads 16: $(COMPEG_SCAN_THREADS=16 python3 tools/e2e_probe.py host 2>&1 | grep -E 'decode_blocking|start_decode' | tr '\n' ' ')"
echo "threads 12: $(COMPEG_SCAN_THREADS=12 python3 tools/e2e_probe.py host 2>&1 | grep -E 'decode_blocking|start_decode' | tr '\n' ' ')"
echo "threads 16 rounds 4: $(COMPEG_SCAN_ROUNDS=4 COMPEG_SCAN_THREADS=16 python3 tools/e2e_probe.py host 2>&1 | grep -E 'decode_blocking|start_decode' | tr '\n' ' ')"
echo "rest alone 256K: $(COMPEG_REST_ALONE=262144 python3 tools/e2e_probe.py host 2>&1 | grep -E 'decode_blocking|start_decode' | tr '\n' ' ')"
unset COMPEG_LIB
echo "release: $(python3 tools/e2e_probe.py host 2>&1 | grep -E 'decode_blocking|start_decode' | tr '\n' ' ')"
