cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_inflate_spans.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python tools/span_probe.py 170 6 1 text 2>/dev/null | grep "spans on"
python tools/span_probe.py 128 6 1 text 128 2>/dev/null | grep "spans on"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/sp -o sp --output-format csv -- python /root/repo/tools/span_probe.py 170 6 1 text > /dev/null 2>&1; cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/sp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if 'span' in r['Name'] or 'inflate' in r['Name']: print(r['Name'][:40], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
