cd /root/repo
python tools/e2e_inflate_probe.py 1024 1024 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_e2e_inflate.txt
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_inflate.py -x -q -m gpu > gpurun_out/r05_test18.txt 2>&1; grep -E "passed|failed|Error" gpurun_out/r05_test18.txt | tail -3
