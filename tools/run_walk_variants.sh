#!/bin/bash
# on the GPU box: text level 9 and config 3 with every variant in flate_amd/lib/var/lib_*.so (k_lz_walk tuning)
for f in flate_amd/lib/libflate_hip.so flate_amd/lib/var/lib_*.so; do
  case $f in *wkprof*) continue;; esac
  for a in "-l 9" "--config 3"; do
    FLATE_HIP_LIB=$PWD/$f python bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['roofline']['kernels_ms_per_step']
print('$(basename $f .so) [$a]', d['value'], 'walk', k.get('k_lz_walk'), 'links', k.get('k_lz_links'))"
  done
done
