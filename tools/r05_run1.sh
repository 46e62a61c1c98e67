set -x
cd /root/repo
bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_variants1.txt
for v in prof ilvprof; do echo "== $v"; FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_$v.so python tools/parse_probe.py 2048 6 text; done 2>&1 | tee gpurun_out/r05_probe1.txt
