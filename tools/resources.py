#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel of the library as compiled here (gfx950), one line per kernel.
usage: python tools/resources.py [extra hipcc flags...]   (what `make -C flate_amd/csrc resources` prints, tabulated)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-ldl",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", "flate_hip.hip"] + sys.argv[1:]
out = subprocess.run(cmd, cwd=os.path.join(root, "flate_amd", "csrc"), capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"remark: (?:Function )?Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
print("%-44s %5s %5s %5s %8s %8s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for r in sorted(rows, key=lambda r: r["name"]):
    print("%-44s %5d %5d %5d %8d %8d %4d" % (r["name"][:44], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("SGPRs", -1), r.get("ScratchSize", -1), r.get("LDS", -1), r.get("Occupancy", -1)))
