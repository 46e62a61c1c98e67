#!/bin/bash
# usage: tools/pmc_quick.sh <kernel-substring> <bench args...>  -- one PMC pass (instruction counts) + timing, printed per chunk
k=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pq -o p --output-format csv -- python $R/bench.py "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pq /tmp/pq.json
python - "$k" <<'PY'
import json, sys
d = json.load(open('/tmp/pq.json'))
for name in sorted(d):
    if sys.argv[1] in name:
        c = d[name]['counters']; n = d[name]['dispatches'] * 16385
        print(name, {a: round(v / n) for a, v in sorted(c.items())}, 'us', round(d[name]['trace_us']['avg'], 1))
PY
