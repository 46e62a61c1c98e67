#!/usr/bin/env python3
"""Timeline of the LAST pinned host-buffer compress call of tools/e2e_probe.py from a rocprofv3 kernel + memory-copy trace dir."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the pinned calls come first in each sub-batch setting; find the last k_lz_parse burst groups: print the last 140 events
t0 = None
sel = ev[-int(sys.argv[2]) if len(sys.argv) > 2 else -150:]
t0 = sel[0][0]
for a, b, n in sel:
    print("%9.3f %9.3f %8.3f  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, n))
