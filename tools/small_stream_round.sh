cd /root/repo
for lv in 6 9; do for mib in 0.3 1 4 16 48 200; do
  for cfg in "0 -" "- -"; do
    set -- $cfg
    if [ "$1" = "-" ]; then unset FLATE_HIP_STREAM_WINDOWS; else export FLATE_HIP_STREAM_WINDOWS=$1; fi
    unset FLATE_HIP_STREAM_GROUP
    timeout 120 python tools/small_stream_probe.py $mib $lv 2>&1 | grep "MiB"
  done
done; done
for mib in 16 48 100; do for cfg in "0 -" "- -"; do set -- $cfg; if [ "$1" = "-" ]; then unset FLATE_HIP_STREAM_WINDOWS; else export FLATE_HIP_STREAM_WINDOWS=$1; fi; timeout 120 python tools/small_stream_probe.py $mib 9 tar 2>&1 | grep "MiB"; done; done
unset FLATE_HIP_STREAM_WINDOWS
timeout 300 python tools/stream_probe.py 262144 6 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
timeout 300 python tools/stream_probe.py 1024 6 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
timeout 300 python tools/stream_probe.py 16384 6 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
