export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 tools/overlap_probe.py 960x720:60:256 2>&1 | grep -v "amdgpu.ids\|Exception ignored\|Traceback\|File \|AttributeError" ; }
COMPEG_VERBOSE=1 timeout -k 10 200 python3 tools/overlap_probe.py 960x720:60:256 2>&1 | grep "plan" | sort | uniq -c
run A=1
run COMPEG_WALK_WPB=1 COMPEG_WALK_CHUNK=2 COMPEG_WALK_ROWS=24 COMPEG_WPB=6
run COMPEG_WALK_WPB=1 COMPEG_WALK_CHUNK=4 COMPEG_WALK_ROWS=32 COMPEG_WPB=5
run COMPEG_WALK_WPB=1 COMPEG_WALK_CHUNK=2 COMPEG_WALK_ROWS=24 COMPEG_WPB=4
run COMPEG_WALK_WPB=1 COMPEG_WALK_CHUNK=8 COMPEG_WALK_ROWS=48 COMPEG_WPB=4
run COMPEG_WALK_WPB=1 COMPEG_WALK_CHUNK=16 COMPEG_WALK_ROWS=64 COMPEG_WPB=12
run COMPEG_WPB=8
run COMPEG_WPB=6
