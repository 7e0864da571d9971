#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03_k; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for N in 256 512; do
  rm -rf /tmp/tr$N
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$N -- python3 $GRAFT_REPO_ROOT/tools/trace_small_lml.py $N > $O/trace_$N.out 2>/dev/null
  f=$(find /tmp/tr$N -name "*kernel_trace.csv" | head -1)
  echo "== N=$N"; tail -1 $O/trace_$N.out
  python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $f | head -24
done
