#!/bin/bash
# usage: tools/kinds_bench.sh  -- level-6 per-kernel times on text, zeros, the Silesia-like mix, and its segment types
R=$(cd "$(dirname "$0")/.." && pwd)
for w in text zeros silesia; do
  python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['roofline']['kernels_ms_per_step'])"
done
python $R/tools/kind_probe.py
