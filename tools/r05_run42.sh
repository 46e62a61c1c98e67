cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_inflate_spans.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
bash tools/r05_run36.sh | grep "==\|resolve\|rounds"
python tools/member_probe.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -4
python tools/member_probe.py 2>/dev/null | grep "all 128"
python tools/span_probe.py 170 6 1 text 2>/dev/null | grep "spans on"
