# levels 8-9 whole streams through k_lz_links / k_lz_walk<true, true>: builds given as arguments (names of flate_amd/lib/var/lib_*.so; "-" = the product)
cd /root/repo
for v in "$@"; do
  if [ "$v" = "-" ]; then unset FLATE_HIP_LIB; else export FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_$v.so; fi
  for kib in 1024 262144; do
    echo "== build $v, level 9, text, streams of $kib KiB"
    FLATE_HIP_STREAM_WINDOWS=1 timeout 300 python tools/stream_probe.py $kib 9 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
  done
done
