cd /root/repo
for f in flate_amd/lib/var/lib_t384.so flate_amd/lib/var/lib_cap12t.so flate_amd/lib/var/lib_cap12tw.so; do
echo "== $f"
FLATE_HIP_LIB=$PWD/$f timeout 300 python tools/span_probe.py 170 6 1 text 2>/dev/null | grep "spans o"
FLATE_HIP_LIB=$PWD/$f timeout 300 python tools/span_probe.py 128 6 1 text 128 2>/dev/null | grep "spans on"
done | tee gpurun_out/r05_span_cap2.txt
