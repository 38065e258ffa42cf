set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r2_pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r2_pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err || { tail -20 gpurun_out/r2_bench_default.err; exit 1; }
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r2_bench_default.json') if l.startswith('{')][-1])
print({k:j[k] for k in ('value','unit','ms_per_step')}, j['roofline'])
print('single', {k:j['single_frame'][k] for k in ('kernel_ms','device_ms_per_frame','host_end_to_end_ms')})
print('e2e', j.get('end_to_end'))
for c in j.get('other_configs',[]): print(c)
PY
