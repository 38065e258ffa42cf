cd "$GRAFT_REPO_ROOT"
export COMPEG_LIB=$PWD/compeg_amd/libcompeg_hip_lab.so PROBE_REPS=6 PROBE_CHECK=1
C="960x720:60:256 960x720:10:64 960x720:10:96 960x720:16:128 960x720:30:64 960x720:60:512 1000x990:7:40 3840x2160:240:16 3840x2160:16:8 3840x2160:30:16 1920x1080:12:24 1920x1080:4:64:95:1 640x360:45:300"
for w in auto 1 0; do echo "== COMPEG_WALK=$w"; if [ $w = auto ]; then unset COMPEG_WALK; else export COMPEG_WALK=$w; fi; timeout -k 10 400 python3 tools/walk_probe.py $C; done
