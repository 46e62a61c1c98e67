cd /root/repo
for f in flate_amd/lib/var/lib_a*.so; do
echo "== $f"
FLATE_HIP_LIB=$PWD/$f python tools/stream_probe.py 1024 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
FLATE_HIP_LIB=$PWD/$f python tools/stream_probe.py 262144 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
done | tee gpurun_out/r05_stream_seg_sweep.txt
