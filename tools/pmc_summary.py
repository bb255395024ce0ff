#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories.
  python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b ... [--out profiles/x.json] [--filter iso_project]"""
import collections, csv, glob, json, sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
out = None
flt = None
for i, a in enumerate(sys.argv):
    if a == "--out": out = sys.argv[i + 1]
    if a == "--filter": flt = sys.argv[i + 1]
dirs = [d for d in dirs if d not in (out, flt)]
res = collections.defaultdict(dict)
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            name = k.split("(")[0].replace("void ", "").strip()
            res[name][c] = sum(v) / len(v)
            res[name]["launches"] = len(v)
if flt:
    res = {k: v for k, v in res.items() if flt in k}
for k, v in sorted(res.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:16.1f}")
if out:
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
