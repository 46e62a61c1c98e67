cd /root/repo
E2E_PINNED_ONLY=1 python tools/e2e_probe.py 512 768 1024 1536 2048 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_e2e_sweep2.txt
