cd /root/repo
export TMPDIR=/tmp
rm -rf /tmp/tlr
FLATE_HIP_RECT=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlr -- python tools/e2e_pinned_once.py 5 2>/dev/null | grep "pinned calls"
python tools/e2e_timeline.py /tmp/tlr 75 > gpurun_out/r05_timeline_rect0.txt 2>&1
