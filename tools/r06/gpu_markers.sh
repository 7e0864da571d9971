#!/bin/bash
# roctx marker ranges of one cycle: rocprofv3 --marker-trace --kernel-trace, the program directly behind `--`
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GPRY_HIP_ROCTX=1
rm -rf /tmp/mk
timeout 900 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/mk -o p -- python3 $R/tools/r06/prof_markers.py > $R/gpurun_out/r06/markers.log 2>&1
tail -2 $R/gpurun_out/r06/markers.log
ls /tmp/mk/* | head
python3 - <<'PY' | tee $R/gpurun_out/r06/markers_summary.txt
import csv, glob, collections
f = glob.glob("/tmp/mk/**/*marker_api_trace.csv", recursive=True)
k = glob.glob("/tmp/mk/**/*kernel_trace.csv", recursive=True)
print("marker file:", f, "kernel file:", k)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("columns:", list(rows[0].keys()) if rows else None, "rows:", len(rows))
    acc = collections.OrderedDict()
    for r in rows:
        name = r.get("Function") or r.get("Message") or r.get("Name") or "?"
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += dur
    print("range | count | host-side total us | us per range")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]): print(f"{n} | {c} | {t:.0f} | {t / c:.1f}")
if k:
    rows = list(csv.DictReader(open(k[0])))
    print("kernel launches in the trace:", len(rows))
PY
