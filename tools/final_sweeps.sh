cd /root/repo
( timeout 900 python tools/span_sweep.py 31 150 2>/dev/null | tail -1
  timeout 600 python tools/span_sweep.py 32 40 big 2>/dev/null | tail -1
  timeout 600 python tools/span_sweep.py 33 100 many 2>/dev/null | tail -1
  timeout 1800 python tools/parity_sweep.py 91 8 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 1800 python tools/parity_sweep.py 92 4 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=3 timeout 1800 python tools/parity_sweep.py 93 4 2>/dev/null | tail -1 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_sweeps3.txt
