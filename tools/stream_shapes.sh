cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r05_test22.txt 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r05_test22.txt | tail -5
for kib in 262144 32768 8192 2048 1024; do
  for w in 1 0; do echo "== STREAM_WINDOWS=$w"; FLATE_HIP_STREAM_WINDOWS=$w python tools/stream_probe.py $kib 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; done
done | tee gpurun_out/r05_stream_shapes.txt
