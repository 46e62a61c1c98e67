cd /root/repo
bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_variants3.txt
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_newprof.so python tools/parse_probe.py 2048 6 text 2>&1 | tee gpurun_out/r05_probe3.txt
