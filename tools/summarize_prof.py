#!/usr/bin/env python3
"""tools/summarize_prof.py <rocprofv3 out dir> <profiles/name>: copies the kernel_stats CSV and writes a markdown
summary (per-kernel calls, average/min/max duration) next to it."""
import csv, glob, os, re, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
stats = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)[0]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
shutil.copy(stats, dst + "_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
OPS = {1: "rescale", 2: "kick", 4: "drift", 8: "KE", 16: "posDelta", 32: "move"}
def pretty(n):
    m = re.search(r"tile_kernel<(\d), (\d+), (\d)>", n)
    if m:
        ops = int(m.group(2))
        return "tile_kernel<%s, %s>" % (["single", "mixed", "double"][int(m.group(1))], "+".join(v for k, v in OPS.items() if ops & k))
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("tgnh::", "")[:60]
with open(dst + "_summary.md", "w") as f:
    f.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
    for r in rows:
        if float(r["Percentage"]) < 0.05: continue
        f.write(f"| `{pretty(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {int(r['MinNs'])/1e3:.2f} | {int(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
print(open(dst + "_summary.md").read())
