#!/bin/bash
# per-launch durations of the fused Cholesky (last factorisation of tools/prof_factor.py N d 2)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for N in 2048 4096; do
  rm -rf /tmp/tr$N
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$N -o t -- python3 $R/tools/prof_factor.py $N 16 2 > /tmp/tr$N.out 2>/dev/null
  echo "N=$N"; tail -7 /tmp/tr$N.out
  python3 $R/tools/trace_fused.py /tmp/tr$N $((N / 64))
done
