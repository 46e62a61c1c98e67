#!/bin/bash
# rocprofv3 --kernel-trace --stats of config #3 as ONE gzip -9 stream (tools/stream_walk_probe2.sh: the TAR-like buffer, text and
# the Silesia-like mix as whole streams of levels 8-9); summary into gpurun_out/<tag>_l9_streams_kernel_stats.csv
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf /tmp/prof_l9s
rocprofv3 --kernel-trace --stats -d /tmp/prof_l9s -o p --output-format csv -- bash $R/tools/stream_walk_probe2.sh > $R/gpurun_out/${TAG}_l9_streams_probe.txt 2> /tmp/prof_l9s.err
f=$(find /tmp/prof_l9s -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- bash tools/stream_walk_probe2.sh   (whole streams of levels 8-9: TAR-like 169 MiB as one gzip -9 stream, 256 MiB of text as 1 / 256 / 64 streams, 96 MiB as 700, the Silesia-like mix as 1 / 16; four calls each)"; grep -E '^"?Name|^"?(void )?k_[a-z_0-9]+[<(]' "$f" | head -40; } > $R/gpurun_out/${TAG}_l9_streams_kernel_stats.csv
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $R/gpurun_out/${TAG}_l9_streams_probe.txt
