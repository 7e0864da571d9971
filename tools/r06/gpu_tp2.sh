#!/bin/bash
# throughput schedule: per-kernel totals (one stream, so that durations add up), small B, block / tail grid
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
{
for a in "4096 16 1 1" "4096 16 2 2" "4096 16 4 2" "4096 16 8 2" "2048 16 1 1" "2048 16 4 2" "2048 16 8 2" "2048 16 16 2" "1024 8 16 2" "1024 8 4 2"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
for bt in "256 512" "256 1024" "512 512" "768 1024" "1024 1024" "512 2048"; do
  timeout 300 python3 tools/r06/check_tp.py 4096 16 16 2 $bt
  timeout 300 python3 tools/r06/check_tp.py 2048 16 32 2 $bt
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp2.log
cd /tmp && export TMPDIR=/tmp
for shape in "4096 16 16 1" "2048 16 32 1" "1024 8 32 1"; do
  set -- $shape
  rm -rf /tmp/tb
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -o p -- python3 $R/tools/r06/prof_tp.py $1 $2 $3 $4 > $R/gpurun_out/r06/prof_tp_$1_$3.log 2>&1
  f=$(find /tmp/tb -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY' | tee $R/gpurun_out/r06/prof_tp_$1_$3.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
first = max(i for i, c in enumerate(ch) if "scale_train" in c[2])
tot, cnt = {}, {}
for c in ch[first:]:
    tot[c[2]] = tot.get(c[2], 0) + (c[1] - c[0]); cnt[c[2]] = cnt.get(c[2], 0) + 1
span = (max(c[1] for c in ch[first:]) - ch[first][0]) / 1e3
print(f"span of the last call: {span:.1f} us; sum of kernel durations {sum(tot.values()) / 1e3:.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"  {v / 1e3:9.1f} us  x{cnt[k]:4d}  {k}")
PY
done
