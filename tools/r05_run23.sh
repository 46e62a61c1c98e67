cd /root/repo
python tools/dbg_groups.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
