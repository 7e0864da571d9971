#!/bin/bash
# round 6: HBM-side traffic of the dominant kernel of THIS build (VERDICT r05 weak point 5).  rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE
# in separate passes (the program directly behind `--`), sweep of 4 x 32768 candidates at N = 4096, d = 16 (= the chunks of the
# headline) and of 1e5 candidates at N = 1024, d = 8 (BASELINE configs[1]: ONE launch).  Per launch of sweep_gemm_dma_sp_kernel;
# FETCH_SIZE is reported in KB and is doubled for gfx950's 128-byte requests tallied at 64 (MI355X_MICROARCH.md, HBM).
#   tools/r06/pmc_traffic.sh -> gpurun_out/r06/pmc_traffic.txt
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r06/pmc_traffic.txt
for what in "config2:python3 $R/tools/prof_sweep.py 4096 16 131072 32768" "config1:python3 $R/tools/prof_sweep.py 1024 8 100000 0"; do
  tag=${what%%:*}; cmd=${what#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    out=/tmp/pmc_${tag}_$c; rm -rf $out
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o p -- $cmd > $out.log 2>&1
    python3 - "$out" "$tag" "$c" >> $R/gpurun_out/r06/pmc_traffic.txt <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(sys.argv[2], sys.argv[3], "no counters"); sys.exit(0)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    acc[name][0] += 1; acc[name][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if v > 0 and ("sweep" in k or "cross_build" in k):
        print(f"{sys.argv[2]}\t{sys.argv[3]}\t{k[:60]}\tlaunches {n}\tper launch {v / n:.6g} KB")
PY
  done
done
cat $R/gpurun_out/r06/pmc_traffic.txt
