cd /root/repo
for f in flate_amd/lib/var/lib_w*.so; do
echo "== $f"
FLATE_HIP_LIB=$PWD/$f python tools/span_probe.py 170 6 1 text 2>/dev/null | grep "spans on"
FLATE_HIP_LIB=$PWD/$f python tools/span_probe.py 128 6 1 text 128 2>/dev/null | grep "spans on"
FLATE_HIP_LIB=$PWD/$f python tools/span_probe.py 128 6 1 silesia 2>/dev/null | grep "spans on"
done | tee gpurun_out/r05_span_wbits.txt
