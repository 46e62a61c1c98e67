cd /root/repo
for tw in 0 1; do
echo "== FLATE_HIP_SPAN_TWO_RUNS=$tw"
FLATE_HIP_SPAN_TWO_RUNS=$tw python tools/span_probe.py 128 6 1 silesia 2>/dev/null | grep "spans"
done | tee gpurun_out/r05_span_sil.txt
