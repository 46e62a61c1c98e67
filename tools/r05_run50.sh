cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_compress.py tests/test_gpu_stream.py tests/test_gpu_flush.py tests/test_gpu_baseline_shapes.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
for v in 1 0; do echo "== FLATE_HIP_MEMSET_INLINE=$v"
FLATE_HIP_MEMSET_INLINE=$v timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-decompress 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
FLATE_HIP_MEMSET_INLINE=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-decompress 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
