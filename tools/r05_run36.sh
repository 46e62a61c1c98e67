cd /root/repo
for c in rec1M-ours rec1M text1M-l6; do echo "== $c"
FL_PAR_CASE=$c FLATE_HIP_INFLATE_SPANS=0 FL_PAR_PROF=1 FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_parprof.so python tools/par_probe.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 | grep -v "hdr:\|nojoin\|long"
done | tee gpurun_out/r05_par_prof.txt
