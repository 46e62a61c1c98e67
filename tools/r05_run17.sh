cd /root/repo
bash tools/run_variants.sh 2>/dev/null | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_test17.txt 2>&1; grep -E "passed|failed|Error" gpurun_out/r05_test17.txt | tail -3
