cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_flush.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > gpurun_out/r05_test16.txt 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r05_test16.txt | tail -5
python tools/stream_probe.py 1024 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_stream_probe.txt
python tools/stream_probe.py 1024 4 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/r05_stream_probe.txt
python tools/stream_probe.py 256 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/r05_stream_probe.txt
