cd /root/repo
bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_variants_a.txt
