#!/bin/bash
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
{
timeout 1500 python3 -m pytest tests/test_lml_batch_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "riding or timed_out or pipelined or appended" 2>&1 | tail -5
for a in "1024 8 32 2" "2048 16 32 2" "4096 16 16 2" "4096 16 14 2" "4096 16 42 2"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
GPRY_HIP_FIT_SCHEDULE=throughput timeout 600 python3 tools/r06/time_fit.py 4096 16 2
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp4.log
cd /tmp && export TMPDIR=/tmp
for shape in "4096 16 16 1" "4096 16 16 2"; do
  set -- $shape
  rm -rf /tmp/tb
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -o p -- python3 $R/tools/r06/prof_tp.py $1 $2 $3 $4 > $R/gpurun_out/r06/prof_tp4_$1_$3_s$4.log 2>&1
  f=$(find /tmp/tb -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY' | tee $R/gpurun_out/r06/prof_tp4_$1_$3_s$4.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
first = max(i for i, c in enumerate(ch) if "scale_train" in c[2])
# with two streams the last call has two scale_train launches: go back to the one before
if sum(1 for c in ch[first - 200:first] if "scale_train" in c[2]) and "s2" in sys.argv[1]: pass
tot, cnt = {}, {}
for c in ch[first:]:
    tot[c[2]] = tot.get(c[2], 0) + (c[1] - c[0]); cnt[c[2]] = cnt.get(c[2], 0) + 1
span = (max(c[1] for c in ch[first:]) - ch[first][0]) / 1e3
print(f"span from the last scale_train: {span:.1f} us; sum of kernel durations {sum(tot.values()) / 1e3:.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"  {v / 1e3:9.1f} us  x{cnt[k]:4d}  {k}")
PY
done
