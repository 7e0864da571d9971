#!/bin/bash
# per-kernel time of ONE objective evaluation at the large sizes (rocprofv3 --kernel-trace --stats): which launches of the chain sit where
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "8192 20 1" "4096 16 1"; do
  set -- $shape
  rm -rf /tmp/profs_$1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$1 -o p -- python3 $R/tools/r04/prof_lml_batch.py $1 $2 $3 10 > $R/gpurun_out/r04/prof_single_$1.log 2>&1
  f=$(find /tmp/profs_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r04/prof_single_$1_kernel_stats.csv
  f=$(find /tmp/profs_$1 -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $R/gpurun_out/r04/prof_single_$1_last_eval.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last evaluation: from the last kernel_train_q launch on
idx = max(i for i, r in enumerate(rows) if "kernel_train_q" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  grid {r['Grid_Size_X']:>8s} x{r['Grid_Size_Y']} x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']:>4s} lds {r['LDS_Block_Size']:>7s} q{r['Queue_Id']} {r['Kernel_Name'][:90]}")
PY
done
cd $R && timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined or cholesky or factor" 2>&1 | tail -5
