cd /root/repo
( timeout 900 python tools/span_sweep.py 21 120 2>/dev/null | tail -1
  timeout 600 python tools/span_sweep.py 22 40 big 2>/dev/null | tail -1
  timeout 600 python tools/span_sweep.py 23 80 many 2>/dev/null | tail -1
  timeout 1500 python tools/parity_sweep.py 78 6 2>/dev/null | tail -2
  timeout 600 python tools/inflate_fuzz.py 2>/dev/null | tail -2 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_sweeps2.txt
