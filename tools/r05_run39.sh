cd /root/repo
( timeout 900 python tools/span_sweep.py 11 60 2>/dev/null | tail -3
  timeout 600 python tools/span_sweep.py 12 20 big 2>/dev/null | tail -3
  timeout 600 python tools/span_sweep.py 13 40 many 2>/dev/null | tail -3
  timeout 1200 python tools/parity_sweep.py 77 3 2>/dev/null | tail -4 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_sweeps.txt
