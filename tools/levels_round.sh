#!/bin/bash
# usage: tools/levels_round.sh <tag>  (GPU box)  the headline workload at every level, plus gzip level 6 and zlib level 4:
# MB/s, ms per step, kernel ms -> gpurun_out/<tag>_levels.txt (first line = the command)
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}
O=$R/gpurun_out/${TAG}_levels.txt
echo "# for l in 4..9: python bench.py -l \$l --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress ; then -g (gzip, level 6) and -z -l 4" > $O
run() {
  python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $O
}
for l in 4 5 6 7 8 9; do run -l $l; done
run -g
run -z -l 4
cat $O
