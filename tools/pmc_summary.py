#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection / kernel_trace CSV into a short per-kernel summary
(only this repo's k_* kernels), so that the result fits gpurun's 64 MiB return limit."""
import collections
import csv
import glob
import json
import sys


def kname(full):
    k = full.split("(")[0]
    return k[5:] if k.startswith("void ") else k   # template instances: "void k_x<...>"


def main(d, out):
    res = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        meta = {}
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
            meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                             "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        for k in agg:
            res.setdefault(k, {})["counters"] = {c: v for c, v in agg[k].items()}
            res[k]["dispatches"] = max(n[(k, c)] for c in agg[k])
            res[k]["meta"] = meta[k]
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k.startswith("k_"):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            res.setdefault(k, {})["trace_us"] = {"calls": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
