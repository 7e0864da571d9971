#!/bin/bash
# the launch chain of one batched LML + gradient call (N = $1, d = $2, B = $3): every kernel of the last call with its duration and the gap in front
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tc -o p -- python3 $R/tools/r04/prof_lml_batch.py $1 $2 $3 3 > /dev/null 2>&1
python3 - "$(find /tmp/tc -name '*kernel_trace.csv' | head -1)" <<'PY' | tee $R/gpurun_out/r05/trace_chain_$1_$3.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in rows]
first = len(ch) - 1
while first > 0 and "scale_train" not in ch[first][2]: first -= 1
prev = None; tot = {}; gaps = 0
for s, e, n in ch[first:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    gaps += max(gap, 0)
    tot.setdefault(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
    prev = e
print(f"span {(ch[-1][1] - ch[first][0]) / 1e3:.1f} us, {len(ch) - first} launches, gaps {gaps:.1f} us")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print(f"  {n:44s} x{c:3d} {t:8.1f} us")
PY
