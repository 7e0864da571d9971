#!/bin/bash
# PMC counters of the factor chain of THIS build (one counter per pass, --kernel-trace only): matrix-pipe busy cycles and
# executed FP64 MFMA ops per kernel, single evaluation (N = 4096) and a batch of 32 thetas (N = 1024).
#   tools/r05/pmc_factor.sh        -> gpurun_out/r05/pmc_factor.txt
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r05/pmc_factor.txt
for what in "single:python3 $R/tools/prof_factor.py 4096 16 3" "batch32:python3 $R/tools/r04/prof_lml_batch.py 1024 8 32 3"; do
  tag=${what%%:*}; cmd=${what#*:}
  for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES; do
    out=/tmp/pmc_${tag}_$c; rm -rf $out
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o p -- $cmd > $out.log 2>&1
    python3 - "$out" "$tag" "$c" >> $R/gpurun_out/r05/pmc_factor.txt <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(sys.argv[2], sys.argv[3], "no counters"); sys.exit(0)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    acc[name][0] += 1; acc[name][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if v > 0: print(f"{sys.argv[2]}\t{sys.argv[3]}\t{k[:70]}\t{n}\t{v:.6g}")
PY
  done
done
cat $R/gpurun_out/r05/pmc_factor.txt | head -80
