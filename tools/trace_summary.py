#!/usr/bin/env python3
"""Per-kernel totals of the LAST evaluation in a rocprofv3 kernel_trace.csv (from the last scale_train launch on),
and for the SYRK trailing updates of the separate-launch Cholesky their durations in order."""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("scale_train")]
lo = starts[-2] if len(starts) >= 2 else 0
hi = starts[-1] if len(starts) >= 2 else len(rows)
seg = rows[lo:hi]
tot = collections.OrderedDict()
for r in seg:
    n = r["Kernel_Name"][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = tot.setdefault(n, [0, 0.0])
    c[0] += 1; c[1] += d
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"one evaluation: {len(seg)} kernels, {span:.0f} us from first start to last end")
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us  {c:4d} x  {n}")
syrk = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if "gemm_dma_kernel<false, true, 2>" in r["Kernel_Name"]]
if syrk:
    print("trailing updates in order (us):", " ".join(f"{v:.0f}" for v in syrk))
pan = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if r["Kernel_Name"].startswith("chol_panel_kernel")]
if pan:
    print("panel steps in order (us):", " ".join(f"{v:.0f}" for v in pan))
