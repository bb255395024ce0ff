#!/usr/bin/env python3
"""Prints the kernels around the LAST launch of a named kernel in a rocprofv3 kernel_trace.csv (start / end in ms
relative to it): python tools/trace_window.py trace.csv pv_max [before] [after]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
after = int(sys.argv[4]) if len(sys.argv) > 4 else 25
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(name)][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[max(0, idx - before):idx + after]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:9.3f} {e:9.3f} {e - s:8.3f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:60]}")
