#!/bin/bash
# on the GPU box: for every tuning build flate_amd/lib/var/lib_*.so (tools/build_variants.sh) the headline's kernel times
# (bench.py, 3 timed steps: sampled chunks against the oracle and the round trip through inflate are part of the run) and the
# tokenizer's parity tests through that build.  usage: tools/variant_round.sh [out file] [names...]
out=${1:-gpurun_out/r06/variants.txt}; shift
mkdir -p $(dirname $out)
names="$@"
[ -z "$names" ] && names=$(ls flate_amd/lib/var/lib_*.so | xargs -n1 basename | sed 's/^lib_//; s/\.so$//' | grep -v '^prof')
for n in $names; do
  f=$PWD/flate_amd/lib/var/lib_$n.so
  FLATE_HIP_LIB=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$n', d['value'], 'roundtrip', d['decompress']['roundtrip_equal'] if d.get('decompress') else None, {k: round(v, 3) for k, v in d['roofline']['kernels_ms_per_step'].items()})" >> $out 2>&1
  if [ -z "$NO_TESTS" ]; then
    FLATE_HIP_LIB=$f python -m pytest -q -x -m gpu tests/test_gpu_compress.py -k "tokenizer_matches_oracle or bytes_match_oracle or runny_inputs" 2>&1 | tail -1 >> $out
  fi
done
cat $out
