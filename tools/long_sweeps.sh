cd /root/repo
mkdir -p gpurun_out/r06
( timeout 1700 python tools/parity_sweep.py ${SEED:-1001} 60 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 timeout 1200 python tools/parity_sweep.py $((${SEED:-1001}+1)) 30 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=1 timeout 900 python tools/parity_sweep.py $((${SEED:-1001}+2)) 20 2>/dev/null | tail -1
  timeout 900 python tools/flush_sweep.py $((${SEED:-1001}+3)) 600 2>/dev/null | tail -1
  FLATE_HIP_STREAM_WINDOWS=1 FLATE_HIP_STREAM_GROUP=2 timeout 900 python tools/flush_sweep.py $((${SEED:-1001}+4)) 400 2>/dev/null | tail -1
  timeout 900 python tools/threshold_sweep.py $((${SEED:-1001}+5)) 400 2>/dev/null | tail -1
  timeout 900 python tools/edge_sweep.py $((${SEED:-1001}+6)) 300 2>/dev/null | tail -1
  timeout 900 python tools/edge_sweep.py $((${SEED:-1001}+7)) 500 chunk 2>/dev/null | tail -1
  timeout 900 python tools/depth_sweep.py $((${SEED:-1001}+8)) 150 2>/dev/null | tail -1
  timeout 1200 python tools/inflate_fuzz.py $((${SEED:-1001}+9)) 30 2>/dev/null | tail -1
  timeout 900 python tools/span_sweep.py $((${SEED:-1001}+10)) 200 2>/dev/null | tail -1
  timeout 900 python tools/span_sweep.py $((${SEED:-1001}+11)) 60 big 2>/dev/null | tail -1 ) | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/long_sweeps.txt
