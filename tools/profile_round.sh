#!/bin/bash
# usage: tools/profile_round.sh <tag>  -- rocprofv3 --kernel-trace --stats of the bench configurations
# (run on the GPU box through gpurun); per-kernel summaries land in gpurun_out/<tag>_*_kernel_stats.csv
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
run() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p --output-format csv -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_${name}_bench.json 2> /tmp/prof_$name.err
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py $*"; grep -E '^"?Name|^"?(void )?k_[a-z_0-9]+[<(]' "$f" | head -40; } > $R/gpurun_out/${TAG}_${name}_kernel_stats.csv
  tail -1 $R/gpurun_out/${TAG}_${name}_bench.json | cut -c1-400
}
run config2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras
run config3 --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras
run config4 --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-extras
run config5 --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras
# one long stream through the inflater (spans): the kernels of tools/span_probe.py
rm -rf /tmp/prof_onestream
rocprofv3 --kernel-trace --stats -d /tmp/prof_onestream -o p --output-format csv -- python $R/tools/span_probe.py 170 6 1 text > $R/gpurun_out/${TAG}_onestream_probe.txt 2> /tmp/prof_onestream.err
f=$(find /tmp/prof_onestream -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/span_probe.py 170 6 1 text   (one 170 MiB gzip level-6 stream: made by the whole-stream compressor, inflated once the old way and once by spans)"; grep -E '^"?Name|^"?(void )?k_[a-z_0-9]+[<(]' "$f" | head -40; } > $R/gpurun_out/${TAG}_onestream_kernel_stats.csv
