cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_inflate_spans.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for tw in 450 500 540 565 600 650; do
echo "== FLATE_HIP_SPAN_TWIN=$tw"
FLATE_HIP_SPAN_TWIN=$tw python tools/span_probe.py 128 6 1 text 128 2>/dev/null | grep "spans on"
done | tee gpurun_out/r05_span_twin_sweep.txt
FLATE_HIP_SPAN_DEBUG=1 python tools/span_probe.py 170 6 1 text 2>&1 | grep "spans\]" | tail -12
