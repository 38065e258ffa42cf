set -e
timeout -k 10 200 python tools/coop_stress.py 4 > gpurun_out/team_stress.log 2>&1 || { tail -5 gpurun_out/team_stress.log; exit 1; }
grep bad gpurun_out/team_stress.log
for sz in "3840 2160" "1920 1080" "1280 720"; do set -- $sz
COMPEG_COOP=1 COMPEG_LIB=gpurun_ab/lib_coopstamps.so timeout -k 10 120 python tools/coop_stamps_probe.py $1 $2 4 2>&1 | head -5
COMPEG_COOP=1 timeout -k 10 120 python bench.py --width $1 --height $2 --batch 2 --steps 3 --cpu-seconds 0 --no-extra-configs --e2e-reps 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['single_frame']; print('$1x$2 kernel_ms', j['kernel_ms'], 'device_ms', j['device_ms_per_frame'], 'host e2e', j['host_end_to_end_ms'])
"
done
