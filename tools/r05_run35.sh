cd /root/repo
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python bench.py > gpurun_out/r05_bench_mid.json 2> gpurun_out/r05_bench_mid.err; tail -c 300 gpurun_out/r05_bench_mid.json
