#!/bin/bash
# usage: tools/traffic_round.sh <tag>   (on the GPU box)  HBM traffic of the headline bench per kernel:
# FETCH_SIZE and WRITE_SIZE need separate rocprofv3 --pmc passes (TCC counter slots, MI355X_MICROARCH.md);
# result: gpurun_out/<tag>_hbm_traffic_1gib.json in the layout bench.py reads from profiles/.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-verify"
rm -rf /tmp/tf /tmp/tw
rocprofv3 --pmc FETCH_SIZE -d /tmp/tf -o p --output-format csv -- python $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/tw -o p --output-format csv -- python $R/bench.py $ARGS > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/tf /tmp/tf.json
python $R/tools/pmc_summary.py /tmp/tw /tmp/tw.json
python - "$R/gpurun_out/${TAG}_hbm_traffic_1gib.json" <<'PY'
import json, sys
f = json.load(open('/tmp/tf.json')); w = json.load(open('/tmp/tw.json'))
def per_step(d, k, c):  # the command runs ONE step: everything the kernel's launches of that step moved
    return int(d[k]["counters"][c] * 1024) if k in d and c in d[k].get("counters", {}) else 0
ks = {}
for k in sorted(set(f) | set(w)):
    name = k.split("<")[0]  # (template instantiations of one kernel -- k_lz_match plain / runny windows -- add up)
    e = ks.setdefault(name, {"fetch_bytes": 0, "write_bytes": 0})
    e["fetch_bytes"] += per_step(f, k, "FETCH_SIZE")
    e["write_bytes"] += per_step(w, k, "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes of `python bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline --no-extras --no-verify` (1 GiB text, level 6, 16385 chunks, plus the round-trip inflate). "
               "Per bench step (all launches of a kernel in that step added up), counter units of 1 KiB converted to bytes. gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
               "under-reports wide coalesced reads by 2x; these kernels mix narrow gathers and wide loads, so the read "
               "figure is a lower bound between 1x and 2x. Algorithmic bytes of the compress path: n_in + n_out = 1.50e9.",
       "workload": "text", "bytes_per_gpu": 1073741824, "mode": 6, "kernels": ks}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
tot = lambda key: sum(v[key] or 0 for n, v in ks.items() if n.startswith("k_") and n not in ("k_inflate", "k_inflate_par", "k_gather_copy", "k_scan_lens"))
print("compress path: fetch %.2f GB write %.2f GB" % (tot("fetch_bytes") / 1e9, tot("write_bytes") / 1e9))
PY
