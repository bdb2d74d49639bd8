#!/usr/bin/env python3
"""Prints VGPR/SGPR/LDS/occupancy per gfx950 kernel of tgnh_kernels.hip (hipcc -Rpass-analysis)."""
import re, subprocess, sys, os
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openmm_drudenose_amd", "csrc", "tgnh_kernels.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-x", "hip", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur: rows.append(cur)
        cur = {"name": subprocess.run(["c++filt", t.split(":",1)[1].strip()], capture_output=True, text=True).stdout.strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("void tgnh::", "")
    print(f"{n:55s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('SGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
