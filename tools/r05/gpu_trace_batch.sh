#!/bin/bash
# per-launch durations of the batched panel steps (rocprofv3 --kernel-trace), N = 1024, 32 thetas; flags from $1
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GPRY_PANEL_FLAGS=${1:-0}
rm -rf /tmp/tb
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -o p -- python3 $R/tools/r04/prof_lml_batch.py 1024 8 ${2:-32} 3 > $R/gpurun_out/r05/trace_batch_$1.log 2>&1
f=$(find /tmp/tb -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/r05/trace_batch_steps_$GPRY_PANEL_FLAGS.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Z", "")) for r in rows]
# last call: the last 16 chol launches
idx = [i for i, c in enumerate(ch) if "chol_fused" in c[2]]
last = idx[-16:]
prev_end = None
for i in last:
    s, e, n, gx, gz = ch[i]
    gap = (s - prev_end) if prev_end else 0
    print(f"{n} grid=({gx},{gz}) dur={(e - s) / 1e3:.1f} us gap_before={gap / 1e3:.1f} us")
    prev_end = e
print("sum chol:", sum(ch[i][1] - ch[i][0] for i in last) / 1e3, "us")
# whole last call: from the first kernel after the previous call's last
first = last[0]
while first > 0 and "scale_train" not in ch[first][2]: first -= 1
tot = {}
for c in ch[first:]:
    tot[c[2]] = tot.get(c[2], 0) + (c[1] - c[0])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"  {k}: {v / 1e3:.1f} us")
print("span:", (ch[-1][1] - ch[first][0]) / 1e3, "us")
PY
