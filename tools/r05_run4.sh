cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_compress.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05_test4.txt
bash tools/host_path_round.sh > gpurun_out/r05_host_path.txt 2>&1
tail -30 gpurun_out/r05_host_path.txt
