#!/usr/bin/env python3
"""Compile one HIP file for gfx950 and print per-kernel register / scratch / LDS usage (one line each).
usage: python tools/kres.py path/to/file.hip [extra hipcc flags]"""
import re, subprocess, sys, os
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", src, "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split()[0]] = int(m.group(1))
    if "error" in line:
        print(line)
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    print(f"{n:<60} vgpr {r.get('VGPRs',0):>4} agpr {r.get('AGPRs',0):>4} sgpr {r.get('SGPRs',0):>4} scratch {r.get('ScratchSize',0):>5} occ {r.get('Occupancy',0)} lds {r.get('LDS',0)}")
