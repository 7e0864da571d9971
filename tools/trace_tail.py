#!/usr/bin/env python3
"""Print the last n kernels of a rocprofv3 kernel_trace.csv with durations and the gaps between them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {r['Kernel_Name'][:80]}")
    prev_end = e
