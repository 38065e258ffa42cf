#!/bin/bash
# Reproduces round 3's unexplained finding (profiles/r03/NOTES.md): the streamed windows staged by LDS-DMA
# (global_load_lds_dword) decode a few images in a thousand wrong in dense streams; staged through registers (what ships)
# none.  Builds the laboratory library with the LDS-DMA arm (-DCG_STREAM_LDSDMA: compiled out of every other build) HERE
# (hipcc cross-compiles), then -- on the GPU box, through gpurun -- decodes batches like batch 235 of round 3's fuzz seed 9902
# (1463 slots of four 640x360 frames, DRI = 7, 5-8 bit per pixel, the streamed-window kernel forced: tools/repro_ldsdma.py)
# with both libraries and prints the counts of wrong outputs.
#   tools/repro_ldsdma.sh build     (in the container)
#   gpurun -- tools/repro_ldsdma.sh run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  "$ROOT/tools/build_variant.sh" ldsdma "-DCG_STREAM_LDSDMA"
  "$ROOT/tools/build_variant.sh" regs ""
  ls -la "$ROOT"/gpurun_ab/lib_ldsdma.so "$ROOT"/gpurun_ab/lib_regs.so
  exit 0
fi
cd "${GRAFT_REPO_ROOT:-$ROOT}"
for lib in ldsdma regs; do
  COMPEG_LIB=$PWD/gpurun_ab/lib_$lib.so COMPEG_STREAM=1 timeout -k 10 500 python3 tools/repro_ldsdma.py 2>&1 | sed "s/^/$lib: /"
done
