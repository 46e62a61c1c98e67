cd /root/repo
python tools/dbg_stream_sil.py 48 64 96 128 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r05_dbg_sil.txt
