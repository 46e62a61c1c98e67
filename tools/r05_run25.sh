cd /root/repo
timeout 600 bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_variants_a.txt
