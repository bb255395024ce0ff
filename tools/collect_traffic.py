#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in separate runs as the
MI355X guide prescribes) into profiles/<tag>_traffic.json: per-kernel average KB per launch.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/profile_modes.py --modes sdf --reps 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/profile_modes.py --modes sdf --reps 2
  python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_traffic.json
"""
import collections, csv, glob, json, sys

fetch_dir, write_dir, out = sys.argv[1:4]
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 3   # calls of the path per profiled process (profile_modes.py --reps 2: 1 + 2)
res = collections.defaultdict(dict)
for d in (fetch_dir, write_dir):
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            res[k][c] = sum(v) / len(v)
            res[k]["launches_" + c] = len(v)
short = {}
for k, v in res.items():
    name = k.split("(")[0].replace("void ", "").strip()
    short[name] = {"FETCH_SIZE_KB": v.get("FETCH_SIZE"), "WRITE_SIZE_KB": v.get("WRITE_SIZE"),
                   "launches": v.get("launches_FETCH_SIZE", v.get("launches_WRITE_SIZE")),
                   "launches_per_step": (v.get("launches_FETCH_SIZE", v.get("launches_WRITE_SIZE")) or 0) / runs}
step = sum(((k["FETCH_SIZE_KB"] or 0.0) + (k["WRITE_SIZE_KB"] or 0.0)) * k["launches_per_step"] for k in short.values()) * 1024.0
json.dump({"step_bytes": step, "runs": runs, "unit": "KB (1024 B) per launch, average; FETCH_SIZE uncorrected (gfx950 reports 1/2 of wide "
                   "coalesced streaming reads; other access patterns uncalibrated)", "kernels": short},
          open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(short), "kernels")
