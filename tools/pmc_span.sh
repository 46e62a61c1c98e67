#!/bin/bash
# usage: tools/pmc_span.sh <tag>   (GPU box)  issue-side counters of k_inflate_span on one 170 MiB gzip level-6 stream
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH"
i=0
for C in "$P1" "$P2" "$P3"; do
  i=$((i + 1)); rm -rf /tmp/ps$i
  rocprofv3 --pmc $C --kernel-trace -d /tmp/ps$i -o p --output-format csv -- python $R/tools/span_probe.py 170 6 1 text > /tmp/ps$i.out 2> /tmp/ps$i.err
  python $R/tools/pmc_summary.py /tmp/ps$i /tmp/ps$i.json
done
python - "$R/gpurun_out/${TAG}_span_pmc.json" <<'PY'
import json, sys
out = {"command": "rocprofv3 --pmc <pass counters> --kernel-trace -- python tools/span_probe.py 170 6 1 text", "kernels": {}}
for i in (1, 2, 3):
    d = json.load(open('/tmp/ps%d.json' % i))
    for k, v in d.items():
        if "span" not in k: continue
        e = out["kernels"].setdefault(k, {"counters": {}})
        n = max(1, v.get("dispatches", 1))
        for c, x in v.get("counters", {}).items(): e["counters"][c] = x / n
        if "trace_us" in v: e["trace_us_pass%d" % i] = v["trace_us"]["avg"]; e["calls"] = v["trace_us"]["calls"]
        e["meta"] = v.get("meta")
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k in sorted(out["kernels"]):
    c = out["kernels"][k]["counters"]
    print(k, {x: "%.3g" % c[x] for x in sorted(c)}, out["kernels"][k].get("trace_us_pass1"), out["kernels"][k].get("meta"))
PY
