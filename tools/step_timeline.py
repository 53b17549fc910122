#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace CSV as a timeline: start (us from the step's first kernel),
duration, and the gap since the end of whatever ended last (a negative gap: it ran beside something).
    python tools/step_timeline.py <kernel_trace.csv> [pack_kernel]   # the step = from the last-but-one launch of the
                                                                     # anchor kernel to the last one"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "pack_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
last_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0][:52]
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - last_end) / 1e3:7.1f}  {name}")
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
print(f"step (anchor to anchor): {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us; some kernel running: {busy / 1e3:.1f} us")
