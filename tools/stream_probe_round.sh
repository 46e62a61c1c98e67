# whole-stream compress of 256 MiB of text as streams of 1 MiB / 16 MiB / one stream, levels 4, 6, 9 (tools/stream_probe.py); single small streams (tools/small_stream_probe.py)
cd /root/repo
for lv in 4 6 9; do for kib in 1024 16384 262144; do timeout 300 python tools/stream_probe.py $kib $lv 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; done; done
for lv in 6 9; do for mib in 0.3 1 4 16 48 200; do for w in 0 -; do if [ "$w" = "-" ]; then unset FLATE_HIP_STREAM_WINDOWS; else export FLATE_HIP_STREAM_WINDOWS=$w; fi; timeout 120 python tools/small_stream_probe.py $mib $lv 2>&1 | grep "MiB"; done; done; done
