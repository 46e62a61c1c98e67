cd /root/repo
for v in $VARIANTS; do
echo "== $v"; FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_$v.so timeout 1200 python -m pytest tests/test_gpu_compress.py tests/test_gpu_stream.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
done
