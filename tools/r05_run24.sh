cd /root/repo
FLATE_HIP_LIB=flate_amd/lib/var/libflate_hip_prof.so python tools/parse_probe.py 2048 6 text 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_prof_events.txt
