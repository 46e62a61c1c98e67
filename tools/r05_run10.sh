cd /root/repo
export TMPDIR=/tmp
for i in 1 2 3; do
  rm -rf /tmp/tl$i
  rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d /tmp/tl$i -- python tools/e2e_pinned_once.py 5 2>/dev/null | grep "pinned calls" | tee -a gpurun_out/r05_tl_times3.txt
  python tools/e2e_timeline.py /tmp/tl$i 100000 > gpurun_out/r05_timeline_full_$i.txt 2>&1
  ls /tmp/tl$i/*/ | head
  f=$(ls /tmp/tl$i/*/*hip_api_trace.csv 2>/dev/null | head -1); if [ -n "$f" ]; then python - "$f" > gpurun_out/r05_hipapi_$i.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# long host calls only
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if b - a > 200000:
        print("%10.3f %8.3f %s" % ((a - t0) / 1e6, (b - a) / 1e6, r["Function"]))
PY
  fi
done
