#!/bin/bash
# PMC counters of cross_build_kernel (one counter group per pass, --kernel-trace only).  usage: tools/pmc_cross_build.sh
export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  out=gpurun_out/pmc_cb
  rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o t -- python3 tools/prof_sweep.py 4096 16 131072 32768 > $out/out.log 2>&1
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counters", open(sys.argv[1] + "/out.log").read()[-400:]); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "cross_build_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k}: {sum(v) / len(v):.4g} per launch ({len(v)} launches)")
PY
done
