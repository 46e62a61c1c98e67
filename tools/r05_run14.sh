cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_compress.py tests/test_gpu_flush.py -x -q -m gpu > gpurun_out/r05_test14.txt 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r05_test14.txt | tail -5
python tools/stream_probe.py 1024 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_stream_probe.txt
FLATE_HIP_STREAM_WINDOWS=0 python tools/stream_probe.py 1024 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/r05_stream_probe.txt
bash tools/run_variants.sh 2>/dev/null | tail -3
