cd /root/repo
timeout 900 python -m pytest tests/test_gpu_compress.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r05_test2.txt
bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_variants2.txt
