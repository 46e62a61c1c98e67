#!/bin/bash
# usage: tools/pmc_parse.sh <tag> [bench args...]   (on the GPU box)
# Is the LDS array the bound of k_lz_parse?  Three rocprofv3 --pmc passes (8 SQ slots each, counters only: no
# trace domains beside --kernel-trace) of one headline bench step; per-kernel sums land in
# gpurun_out/<tag>_parse_pmc.json with the command as the first entry.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}; shift
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-verify --no-decompress $*"
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
P3="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES"
i=0
for C in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  rm -rf /tmp/pp$i
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pp$i -o p --output-format csv -- python $R/bench.py $ARGS > /tmp/pp$i.out 2> /tmp/pp$i.err
  python $R/tools/pmc_summary.py /tmp/pp$i /tmp/pp$i.json
done
python - "$R/gpurun_out/${TAG}_parse_pmc.json" "$ARGS" <<'PY'
import json, sys
out = {"command": "rocprofv3 --pmc <pass counters> --kernel-trace -- python bench.py " + sys.argv[2],
       "note": "three separate passes; values are sums over all waves / SEs of ONE launch (1 GiB text, 16385 chunks); "
               "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)", "kernels": {}}
for i in (1, 2, 3):
    d = json.load(open('/tmp/pp%d.json' % i))
    for k, v in d.items():
        e = out["kernels"].setdefault(k, {"counters": {}})
        n = max(1, v.get("dispatches", 1))
        for c, x in v.get("counters", {}).items():
            e["counters"][c] = x / n
        if "trace_us" in v:
            e["trace_us_pass%d" % i] = v["trace_us"]["avg"]
        e["meta"] = v.get("meta")
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k in sorted(out["kernels"]):
    if "parse" in k or "chain" in k or "emit" in k:
        print(k, json.dumps(out["kernels"][k], sort_keys=True))
PY
