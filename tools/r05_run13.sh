cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_test13.txt 2>&1; grep -E "passed|failed" gpurun_out/r05_test13.txt
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_default.json').readline())
print(d['value'], d['ms_per_step'], d['decompress']['value'])
print(json.dumps(d['e2e_host'], indent=0)[:900])
for k, v in d['other_workloads'].items():
    print(k, v.get('MBps'), v.get('ms'))
PY
