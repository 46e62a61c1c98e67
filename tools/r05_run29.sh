cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_inflate_spans.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for tw in 0 1; do
echo "== FLATE_HIP_SPAN_TWO_RUNS=$tw"
FLATE_HIP_SPAN_TWO_RUNS=$tw python tools/span_probe.py 170 6 1 text 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | grep "spans on"
FLATE_HIP_SPAN_TWO_RUNS=$tw python tools/span_probe.py 128 6 1 text 128 2>/dev/null | grep "spans on"
FLATE_HIP_SPAN_TWO_RUNS=$tw python tools/span_probe.py 128 1 0 silesia 2>/dev/null | grep "spans on"
done | tee gpurun_out/r05_span_sym.txt
