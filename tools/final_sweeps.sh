# the round's closing sweeps (one-off tools, minutes): usage on the GPU box: bash tools/final_sweeps.sh
cd /root/repo
mkdir -p gpurun_out/r06
( timeout 1500 python tools/parity_sweep.py 651 16 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 1500 python tools/parity_sweep.py 652 10 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=3 timeout 1500 python tools/parity_sweep.py 653 8 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=0 timeout 1500 python tools/parity_sweep.py 654 6 2>/dev/null | tail -1
  timeout 600 python tools/runny_sweep.py 655 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 600 python tools/runny_sweep.py 656 2>/dev/null | tail -1
  timeout 900 python tools/big_batch_sweep.py 657 2>/dev/null | tail -1
  timeout 600 python tools/edge_sweep.py 658 80 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 600 python tools/edge_sweep.py 659 80 2>/dev/null | tail -1
  timeout 600 python tools/edge_sweep.py 660 200 chunk 2>/dev/null | tail -1
  timeout 600 python tools/threshold_sweep.py 661 150 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 600 python tools/threshold_sweep.py 662 100 2>/dev/null | tail -1
  timeout 600 python tools/depth_sweep.py 663 40 2>/dev/null | tail -1
  timeout 600 python tools/flush_sweep.py 667 150 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=0 timeout 600 python tools/flush_sweep.py 668 100 2>/dev/null | tail -1
  timeout 600 python tools/inflate_edges.py 2>/dev/null | tail -1
  timeout 900 python tools/inflate_fuzz.py 664 12 2>/dev/null | tail -1
  timeout 600 python tools/span_sweep.py 665 100 2>/dev/null | tail -1
  timeout 600 python tools/span_sweep.py 666 40 many 2>/dev/null | tail -1 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/final_sweeps.txt
