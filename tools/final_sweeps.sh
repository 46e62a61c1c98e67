cd /root/repo
mkdir -p gpurun_out/r06
( timeout 1500 python tools/parity_sweep.py 621 24 2>/dev/null | tail -2
  timeout 900 python tools/inflate_fuzz.py 622 20 2>/dev/null | tail -2
  timeout 600 python tools/span_sweep.py 623 150 2>&1 | tail -2
  timeout 600 python tools/span_sweep.py 624 40 big 2>&1 | tail -1
  timeout 600 python tools/span_sweep.py 625 60 many 2>&1 | tail -1
  timeout 600 python tools/runny_sweep.py 626 2>/dev/null | tail -2
  timeout 900 python tools/big_batch_sweep.py 627 2>/dev/null | tail -2 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/final_sweeps2.txt
