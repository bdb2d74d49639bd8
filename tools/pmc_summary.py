#!/usr/bin/env python3
"""tools/pmc_summary.py <dir with rocprofv3 --pmc csv output>...: per-kernel average of each collected counter."""
import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
OPS = {1: "rescale", 2: "kick", 4: "drift", 8: "KE", 16: "posDelta", 32: "move"}
def pretty(n):
    m = re.search(r"tile_kernel<(\d), (\d+), (\d)>", n)
    if m:
        ops = int(m.group(2))
        return "tile<%s,%s>" % (["single", "mixed", "double"][int(m.group(1))], "+".join(v for k, v in OPS.items() if ops & k))
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("tgnh::", "")[:50]
for k, cs in acc.items():
    if "tgnh" not in k: continue
    print(pretty(k), {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
