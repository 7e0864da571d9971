#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_w; mkdir -p $O
for cfg in "256 1" "256 64" "1024 1" "1024 64"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 tools/r03/trace_predict_std.py $1 $2 > $O/run.log 2>&1
  grep "us per call" $O/run.log
  python3 - $O/t <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seg = rows[-12:]
# one call = from the last cross/kstar kernel start backwards: print the last 8 kernels
for r in rows[-8:]:
    print(f"    dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us  {r['Kernel_Name'][:100]}")
PY
  rm -rf $O/t
done
