#!/usr/bin/env python3
"""Prints VGPR/SGPR/LDS/occupancy per gfx950 kernel of tgnh_kernels.hip and tgnh_gather.hip (hipcc -Rpass-analysis), the VGPR spills, and the
number of instructions in the kernel's ISA that touch the stack (`stackops`: scratch_* / buffer_* -- nothing else here uses
buffer instructions).  A small `scratch` with stackops 0 and no VGPR spill is a slot the register allocator reserved for
spilled scalar registers and never used (they went to VGPR lanes): nothing is stored to memory."""
import re, subprocess, sys, os, tempfile
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openmm_drudenose_amd", "csrc")
out, stack = "", {}
with tempfile.TemporaryDirectory() as tmp:
    for name in ("tgnh_kernels.hip", "tgnh_gather.hip"):
        asm = os.path.join(tmp, name + ".s")
        out += subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip",
                               os.path.join(csrc, name), "-o", asm, "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:],
                              capture_output=True, text=True).stderr
        cur = None
        for line in open(asm):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1); stack[cur] = 0
            elif cur and re.match(r"\s+(scratch_|buffer_)", line):
                stack[cur] += 1
            elif cur and "s_endpgm" in line:
                cur = None
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur: rows.append(cur)
        mangled = t.split(":", 1)[1].strip()
        cur = {"name": subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip(), "stackops": stack.get(mangled, "?")}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("void tgnh::", "")
    print(f"{n:55s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('SGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')} vspill {r.get('VGPRs Spill','?')} stackops {r['stackops']}")
