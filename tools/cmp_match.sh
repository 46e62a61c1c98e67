for args in "--workload text" "--workload zeros" "--workload silesia" "--workload text --chunk 1048576" "--workload text --mode 9 --bytes 67108864" "--workload text --mode 4"; do
for d in 0 2; do
echo "== $args dbg=$d"
FLATE_HIP_DBG=$d python bench.py --bytes 268435456 $args --steps 2 --warmup 1 --no-cpu-baseline --no-decompress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernels_ms_per_step'])"
done; done
