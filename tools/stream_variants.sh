for n in s0 s1 s2 s3 s4; do
  f=$PWD/flate_amd/lib/var/lib_$n.so
  for args in "--chunk 1048576 --bytes 268435456" "--chunk 268435456 --bytes 268435456"; do
  FLATE_HIP_LIB=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$n', '$args', d['value'], {k: round(v, 3) for k, v in d['roofline']['kernels_ms_per_step'].items()})"
  done
done
