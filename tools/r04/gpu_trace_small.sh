#!/bin/bash
# every launch of one LML + gradient evaluation at a few hundred points (durations and gaps): is the chain launch-bound?
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "256 4 1" "256 4 9" "512 6 1"; do
  set -- $shape
  rm -rf /tmp/profx
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/profx -o p -- python3 $R/tools/r04/prof_lml_batch.py $1 $2 $3 20 > /tmp/profx.log 2>&1
  grep "per call" /tmp/profx.log
  f=$(find /tmp/profx -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "kernel_train_q" in r["Kernel_Name"] or "scale_train" in r["Kernel_Name"]]
first = [i for i in idx if "scale_train" in rows[i]["Kernel_Name"]] or idx
a = first[-2]; b = first[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - t0
print(f"  launches {len(seg)}, kernel time {busy/1e3:.1f} us, span first start -> last end {span/1e3:.1f} us, period to next evaluation {(int(rows[b]['Start_Timestamp'])-t0)/1e3:.1f} us")
prev = None
out = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    out.append(f"{r['Kernel_Name'][:28]}:{(e-s)/1e3:.1f}(+{gap:.1f})")
    prev = e
print("  " + " ".join(out))
PY
done
