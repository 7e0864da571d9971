#!/bin/bash
# the general chain at N = 256 / 512 / 1024 under the kernel trace: per-kernel durations and start offsets of the last evaluation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_r; mkdir -p $O
for N in 256 512 1024; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$N -o t -- python3 tools/trace_small_lml.py $N > $O/run_$N.log 2>&1
  python3 - $O/t$N $N > $O/trace_$N.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("scale_train")]
seg = rows[starts[-2]:starts[-1]]
t0 = int(seg[0]["Start_Timestamp"])
print(f"N={sys.argv[2]}: {len(seg)} kernels, {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us first start to last end")
prev_end = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gx = [k for k in r if "Grid" in k and k.upper().endswith("X")]
    wx = [k for k in r if "Workgroup" in k and k.upper().endswith("X")]
    grid = int(r[gx[0]]) // max(int(r[wx[0]]), 1) if gx and wx else -1
    print(f"  +{(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:5.1f}  wgs {grid:5d}  {r['Kernel_Name'][:90]}")
    prev_end = e
PY
  rm -rf $O/t$N
done
cat $O/run_*.log $O/trace_256.txt $O/trace_512.txt $O/trace_1024.txt
