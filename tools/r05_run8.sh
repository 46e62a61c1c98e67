cd /root/repo
export TMPDIR=/tmp
for i in 1 2 3 4; do
  rm -rf /tmp/tl$i
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl$i -- python tools/e2e_pinned_once.py 6 2>/dev/null | grep "pinned calls" | tee -a gpurun_out/r05_tl_times.txt
  python tools/e2e_timeline.py /tmp/tl$i 110 > gpurun_out/r05_timeline_$i.txt 2>&1
done
for i in 1 2 3; do python tools/e2e_pinned_once.py 8 2>/dev/null | grep "pinned calls"; done | tee -a gpurun_out/r05_tl_times.txt
