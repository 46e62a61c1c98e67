cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05_test5.txt
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_config4.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
