cd /root/repo
mkdir -p gpurun_out/r06
( FLATE_HIP_STREAM_WINDOWS=1 timeout 1500 python tools/parity_sweep.py 631 10 2>/dev/null | tail -2
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=3 timeout 1500 python tools/parity_sweep.py 632 10 2>/dev/null | tail -2
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=1 timeout 1500 python tools/parity_sweep.py 633 6 2>/dev/null | tail -2
  FLATE_HIP_STREAM_WINDOWS=1 timeout 600 python tools/runny_sweep.py 634 2>/dev/null | tail -2
  timeout 1500 python tools/parity_sweep.py 635 6 2>/dev/null | tail -2 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/final_sweeps3.txt
