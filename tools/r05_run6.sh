cd /root/repo
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_baseline_shapes.py tests/test_gpu_flush.py -x -q -m gpu > gpurun_out/r05_test6.txt 2>&1; tail -3 gpurun_out/r05_test6.txt
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_config4.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
