cd /root/repo
bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_variants_c.txt
VARIANTS="$VARIANTS" bash tools/r05_run26.sh
