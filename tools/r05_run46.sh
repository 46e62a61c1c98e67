cd /root/repo
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_sw3.so timeout 900 python -m pytest tests/test_gpu_inflate_spans.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
for f in flate_amd/lib/libflate_hip.so flate_amd/lib/var/lib_sw2.so flate_amd/lib/var/lib_sw3.so flate_amd/lib/var/lib_sw4.so; do
echo "== $f"
FLATE_HIP_LIB=$PWD/$f timeout 300 python tools/span_probe.py 170 6 1 text 2>/dev/null | grep "spans o"
FLATE_HIP_LIB=$PWD/$f timeout 300 python tools/member_probe.py 2>/dev/null | grep "member  93\|all 128"
done | tee gpurun_out/r05_span_sweeps.txt
