cd /root/repo
mkdir -p gpurun_out/r06
( FLATE_HIP_STREAM_WINDOWS=1 timeout 1500 python tools/parity_sweep.py 641 10 2>/dev/null | tail -2
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=3 timeout 1500 python tools/parity_sweep.py 642 10 2>/dev/null | tail -2
  timeout 1500 python tools/parity_sweep.py 643 16 2>/dev/null | tail -2
  FLATE_HIP_STREAM_WINDOWS=1 timeout 600 python tools/runny_sweep.py 644 2>/dev/null | tail -2
  timeout 600 python tools/runny_sweep.py 645 2>/dev/null | tail -2
  timeout 900 python tools/big_batch_sweep.py 646 2>/dev/null | tail -2
  timeout 600 python tools/span_sweep.py 647 60 2>&1 | tail -1 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/final_sweeps4.txt
