#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of one csrc file, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).   python tools/kernel_resources.py posterior_step.hip [substring] [-DX=Y ...]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
needle = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
defs = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + R + "/include",
       "-I" + R + "/bayesian_cbf_amd/csrc", "-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage", "-c",
       R + "/bayesian_cbf_amd/csrc/" + src, "-o", "/tmp/_kr.o"] + defs
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = dict(name=re.sub(r"\(.*", "", name).replace("void bcbf::", ""))
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
    if "error" in line:
        print(line)
for r in rows:
    if needle in r["name"]:
        print("%-70s vgpr %3d agpr %3d scratch %4d occ %d sgpr %3d lds %6d" % (r["name"][:70], r.get("vgpr", -1), r.get("agpr", -1),
              r.get("scratch", -1), r.get("occ", -1), r.get("sgpr", -1), r.get("lds", -1)))
