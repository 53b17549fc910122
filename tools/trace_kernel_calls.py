#!/usr/bin/env python3
"""Per-call durations of the kernels whose name contains one of the given words, from a rocprofv3 --kernel-trace CSV:
    python tools/trace_kernel_calls.py <kernel_trace.csv> grouped_candidates gp_refine ...
(the stats CSV gives averages only; a skewed job's long pole is ONE call)"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
words = sys.argv[2:]
by = defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    if any(w in name for w in words):
        short = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "")).split("(")[0]
        by[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in by.items():
    print(f"{name}: {len(v)} calls; last 12 (us): {[round(x) for x in v[-12:]]}")
