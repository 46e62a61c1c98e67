cd /root/repo
for k in sparse records tar; do
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_wkprof.so python tools/walk_load_probe.py $k 9 1024 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
done | tee gpurun_out/r05_walk_probe.txt
