#!/bin/bash
# L2 hit / miss counts of the sweep contraction with the upward and the alternating k walk (separate --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_x; mkdir -p $O
for alt in 0 1; do
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $ctr | tr ' ' '_')
    GPRY_SWEEP_ALTWALK=$alt timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/p_${alt}_$tag -o t -- python3 tools/prof_sweep.py 4096 16 131072 32768 3 3 0 > $O/out_${alt}_$tag.log 2>&1
    python3 - $O/p_${alt}_$tag "$alt $ctr" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print(sys.argv[2], "no counters"); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "sweep_gemm" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("altwalk", sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 3) for k, v in acc.items()}, "(1e6 per launch)")
PY
  done
done 2>&1 | tee $O/l2.log
