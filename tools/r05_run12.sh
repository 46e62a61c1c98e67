cd /root/repo
timeout 900 python -m pytest tests/test_gpu_compress.py -x -q -m gpu > gpurun_out/r05_test12.txt 2>&1; grep -E "passed|failed" gpurun_out/r05_test12.txt
for i in 1 2; do python tools/e2e_pinned_once.py 8 2>/dev/null | grep "pinned calls"; done
for i in 1 2 3; do echo "== e2e_probe run $i (median of six calls after two warm-ups)"; python tools/e2e_probe.py 1024 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; done > gpurun_out/r05_e2e_median.txt 2>&1; cat gpurun_out/r05_e2e_median.txt
