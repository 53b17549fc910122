#!/usr/bin/env python3
"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV: start, duration and gap of every
dispatch between the last-but-one and last launches of a marker kernel (default pack_kernel)."""
import csv, re, sys
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "pack_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[-back], marks[-back + 1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n[:70]
busy, prev_end = 0, t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, dispatches {len(seg)}")
