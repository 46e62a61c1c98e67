#!/bin/bash
# usage: tools/pmc_inflate.sh <tag> [bench args...]   (on the GPU box)
# What bounds k_inflate?  Three rocprofv3 --pmc passes (counters only beside --kernel-trace) of one headline bench step
# with the decompress leg; per-kernel sums of k_inflate land in gpurun_out/<tag>_inflate_pmc.json, the command first.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}; shift
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-verify $*"
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH"
i=0
for C in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  rm -rf /tmp/pi$i
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pi$i -o p --output-format csv -- python $R/bench.py $ARGS > /tmp/pi$i.out 2> /tmp/pi$i.err
  python $R/tools/pmc_summary.py /tmp/pi$i /tmp/pi$i.json
done
python - "$R/gpurun_out/${TAG}_inflate_pmc.json" "$ARGS" <<'PY'
import json, sys
out = {"command": "rocprofv3 --pmc <pass counters> --kernel-trace -- python bench.py " + sys.argv[2],
       "note": "three separate passes; values are sums over all waves / SEs of ONE launch (16385 streams of 65535 bytes of text); "
               "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)", "kernels": {}}
for i in (1, 2, 3):
    d = json.load(open('/tmp/pi%d.json' % i))
    for k, v in d.items():
        if "inflate" not in k:
            continue
        e = out["kernels"].setdefault(k, {"counters": {}})
        n = max(1, v.get("dispatches", 1))
        for c, x in v.get("counters", {}).items():
            e["counters"][c] = x / n
        if "trace_us" in v:
            e["trace_us_pass%d" % i] = v["trace_us"]["avg"]
        e["meta"] = v.get("meta")
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k in sorted(out["kernels"]):
    print(k, json.dumps(out["kernels"][k], sort_keys=True))
PY
