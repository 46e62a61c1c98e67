#!/usr/bin/env python3
"""Print the per-kernel counters of gpurun_out/<tag>_pmc{1,2}.json per chunk (arg 2 = chunks per dispatch)."""
import json, sys
tag = sys.argv[1]; per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for f in ('pmc1', 'pmc2'):
    d = json.load(open('gpurun_out/%s_%s.json' % (tag, f)))
    for k in sorted(d):
        c = d[k].get('counters', {}); n = d[k].get('dispatches', 1) * per
        t = d[k].get('trace_us')
        print(f, k, 'disp', d[k].get('dispatches'), {a: round(v / n) for a, v in sorted(c.items())}, 'us', t and round(t['avg'], 1))
