cd /root/repo
mkdir -p gpurun_out/r06
( timeout 1700 python tools/parity_sweep.py 1001 60 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 1200 python tools/parity_sweep.py 1002 30 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=1 timeout 900 python tools/parity_sweep.py 1003 20 2>/dev/null | tail -1
  timeout 900 python tools/flush_sweep.py 1004 600 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=2 timeout 900 python tools/flush_sweep.py 1005 400 2>/dev/null | tail -1
  timeout 900 python tools/threshold_sweep.py 1006 400 2>/dev/null | tail -1
  timeout 900 python tools/edge_sweep.py 1007 300 2>/dev/null | tail -1
  timeout 900 python tools/edge_sweep.py 1008 500 chunk 2>/dev/null | tail -1
  timeout 900 python tools/depth_sweep.py 1009 150 2>/dev/null | tail -1
  timeout 1200 python tools/inflate_fuzz.py 1010 30 2>/dev/null | tail -1
  timeout 900 python tools/span_sweep.py 1011 200 2>/dev/null | tail -1
  timeout 900 python tools/span_sweep.py 1012 60 big 2>/dev/null | tail -1 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/long_sweeps.txt
