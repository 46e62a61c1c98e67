cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_compress.py tests/test_gpu_baseline_shapes.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
timeout 600 python bench.py -l 9 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernels_ms_per_step'])"
