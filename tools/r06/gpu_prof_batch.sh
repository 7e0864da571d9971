#!/bin/bash
# per-kernel totals of ONE call of the batched objective (rocprofv3 --kernel-trace): shapes "N d B" from the arguments
# usage: gpu_prof_batch.sh tag "4096 16 16" "2048 16 32" ...
tag=$1; shift
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "$@"; do
  set -- $shape
  N=$1; d=$2; B=$3
  rm -rf /tmp/tb
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -o p -- python3 $R/tools/r04/prof_lml_batch.py $N $d $B 3 > $R/gpurun_out/r06/prof_batch_${tag}_${N}_${B}.log 2>&1
  f=$(find /tmp/tb -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY' | tee $R/gpurun_out/r06/prof_batch_${tag}_${N}_${B}.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
# the last call: from the last scale_train launch on
first = max(i for i, c in enumerate(ch) if "scale_train" in c[2])
tot, cnt = {}, {}
for c in ch[first:]:
    tot[c[2]] = tot.get(c[2], 0) + (c[1] - c[0]); cnt[c[2]] = cnt.get(c[2], 0) + 1
span = (max(c[1] for c in ch[first:]) - ch[first][0]) / 1e3
print(f"span of the last call: {span:.1f} us; sum of kernel durations {sum(tot.values()) / 1e3:.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"  {v / 1e3:9.1f} us  x{cnt[k]:4d}  {k}")
PY
  tail -2 $R/gpurun_out/r06/prof_batch_${tag}_${N}_${B}.log
done
