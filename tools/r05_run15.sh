cd /root/repo
for v in s17 s24 s32 s48; do echo "== $v"; FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_$v.so python tools/stream_probe.py 1024 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; done | tee gpurun_out/r05_stream_seg.txt
