cd /root/repo
for i in 1 2 3 4; do python tools/e2e_pinned_once.py 8 2>/dev/null | grep "pinned calls"; done | tee gpurun_out/r05_tl_times2.txt
python tools/e2e_probe.py 1024 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/r05_tl_times2.txt
