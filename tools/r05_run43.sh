cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_compress.py tests/test_gpu_stream.py tests/test_gpu_baseline_shapes.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for lib in flate_amd/lib/libflate_hip.so flate_amd/lib/var/lib_norun.so; do echo "== $lib"
FLATE_HIP_LIB=$PWD/$lib timeout 600 python tools/kind_probe.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8
done | tee gpurun_out/r05_runskip_kinds.txt
