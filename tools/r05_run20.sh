cd /root/repo
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_prof.so python tools/parse_probe.py 2048 6 text 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_probe20.txt
