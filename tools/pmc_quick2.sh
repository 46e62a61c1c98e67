#!/bin/bash
# usage: tools/pmc_quick2.sh <kernel-substring> <bench args...>  -- two PMC passes (instruction mix; where the waves wait), per chunk
k=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  rm -rf /tmp/pq
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR"; else C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH"; fi
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pq -o p --output-format csv -- python $R/bench.py "$@" > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pq /tmp/pq.json
  python - "$k" <<'PY'
import json, sys
d = json.load(open('/tmp/pq.json'))
for name in sorted(d):
    if sys.argv[1] in name:
        c = d[name]['counters']; n = d[name]['dispatches'] * 16385
        print(name, {a: round(v / n) for a, v in sorted(c.items())}, 'us', round(d[name]['trace_us']['avg'], 1))
PY
done
