#!/bin/bash
# build tuning variants of the library: tools/build_variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/../flate_amd/csrc"
mkdir -p ../lib/var
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $f -O3 -std=c++17 -fPIC -shared -Wno-unused-function -w -ldl -o ../lib/var/lib_$n.so flate_hip.hip && echo built $n ) &
done
wait
