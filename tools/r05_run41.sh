cd /root/repo
python tools/member_probe.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_member_probe.txt
