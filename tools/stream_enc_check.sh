#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
for kib in 1024 262144; do
  timeout 300 python tools/stream_probe.py $kib 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
done
