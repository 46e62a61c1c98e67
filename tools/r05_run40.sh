cd /root/repo
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/libflate_hip_prof.so python tools/stream_parse_probe.py 256 1024 6 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_stream_parse_prof.txt
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/libflate_hip_prof.so python tools/parse_probe.py 2048 6 text 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | head -9 | tee -a gpurun_out/r05_stream_parse_prof.txt
