cd /root/repo
bash tools/run_variants.sh 2>/dev/null | tail -3
