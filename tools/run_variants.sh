#!/bin/bash
# on the GPU box: bench every variant in flate_amd/lib/var, print kernel times
for f in flate_amd/lib/var/lib_*.so; do
  n=$(basename $f .so)
  FLATE_HIP_LIB=$PWD/$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$n', d['value'], {k: round(v, 2) for k, v in d['roofline']['kernels_ms_per_step'].items()})"
done
