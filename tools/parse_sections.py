#!/usr/bin/env python3
"""Instructions of k_lz_parse<false> between the markers of a -DPZ_SEC build (PZ_EV(K) -> ';;PZSEC K' in the listing).
usage: tools/parse_sections.py  (builds into /tmp/isa_sec; prints VALU / SALU / LDS / VMEM counts from each marker to the next in
listing order -- the listing order follows the source closely but not exactly: a guide for where the instructions are)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/isa_sec", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-DPZ_SEC", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-ldl", "-save-temps=obj",
                "-o", "/tmp/isa_sec/lib.so", os.path.join(root, "flate_amd/csrc/flate_hip.hip")] + sys.argv[1:], check=True, cwd="/tmp/isa_sec")
s = open("/tmp/isa_sec/flate_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read().splitlines()
a = next(i for i, l in enumerate(s) if l.startswith("_Z10k_lz_parseILb0EE") and l.rstrip().endswith(":") or l.startswith("_Z10k_lz_parseILb0EE") and ": ;" in l)
b = next(i for i, l in enumerate(s) if ".size\t_Z10k_lz_parseILb0EE" in l)
names = {0: "block entered", 2: "hit: measure", 4: "measure iteration", 6: "match improved", 8: "transition part", 10: "WAIT polls", 12: "WAIT: published",
         14: "call ended", 16: "... with a match", 18: "emit", 20: "emit a match", 22: "segment end / meet", 24: "lazy: look further", 26: "START_CALL", 28: "LOAD_CAND"}
cur = "before the first marker"; cnt = {}
order = []
for l in s[a:b]:
    t = l.strip()
    m = re.match(r";;PZSEC (\d+)", t)
    if m:
        cur = names.get(int(m.group(1)), m.group(1)); order.append(cur); continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    k = "valu" if t.startswith("v_") else "salu" if t.startswith("s_") else "lds" if t.startswith("ds_") else "vmem" if t.startswith(("global_", "buffer_", "flat_")) else "other"
    cnt.setdefault(cur, {}).setdefault(k, 0); cnt[cur][k] += 1
for nm in ["before the first marker"] + order:
    c = cnt.get(nm, {})
    print("%-26s VALU %4d  SALU %4d  LDS %3d  VMEM %3d" % (nm, c.get("valu", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0)))
