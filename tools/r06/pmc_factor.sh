#!/bin/bash
# PMC counters of the batched factor chains of THIS build (one counter per pass, --kernel-trace only, the program directly behind
# `--`): matrix-pipe busy cycles and executed FP64 MFMA ops per kernel for
#   batch32_latency   32 thetas at N = 1024, latency schedule (the compact panel step with the inverse factor as appended rows)
#   batch32_tp        the same call in the throughput schedule (one stream, so that kernels do not overlap)
#   batch16_tp        16 thetas at N = 4096, throughput schedule (column blocks of 512 + SYRK engine)
#   tools/r06/pmc_factor.sh        -> gpurun_out/r06/pmc_factor.txt
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r06/pmc_factor.txt
for what in "batch32_latency:python3 $R/tools/r04/prof_lml_batch.py 1024 8 32 3" "batch32_tp:python3 $R/tools/r06/prof_tp.py 1024 8 32 1" "batch16_tp:python3 $R/tools/r06/prof_tp.py 4096 16 16 1"; do
  tag=${what%%:*}; cmd=${what#*:}
  for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64; do
    out=/tmp/pmc_${tag}_$c; rm -rf $out
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o p -- $cmd > $out.log 2>&1
    python3 - "$out" "$tag" "$c" >> $R/gpurun_out/r06/pmc_factor.txt <<'PY'
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
python3 - $R/gpurun_out/r06/pmc_factor.txt <<'PY'
import sys, collections
rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1]) if l.count("\t") == 4]
d = collections.defaultdict(dict)
for tag, c, k, n, v in rows: d[(tag, k)][c] = (int(n), float(v))
print("tag | kernel | launches | flop executed | matrix-pipe busy")
for (tag, k), cc in sorted(d.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cc and "GRBM_GUI_ACTIVE" in cc and cc["GRBM_GUI_ACTIVE"][1] > 0:
        busy = cc["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (1024 * cc["GRBM_GUI_ACTIVE"][1] / 8)
        fl = cc.get("SQ_INSTS_VALU_MFMA_MOPS_F64", (0, 0))[1] * 512
        if fl > 1e8: print(f"{tag} | {k[:60]} | {cc['GRBM_GUI_ACTIVE'][0]} | {fl:.3g} | {100 * busy:.1f} %")
PY
