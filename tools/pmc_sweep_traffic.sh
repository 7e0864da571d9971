#!/bin/bash
# FETCH_SIZE (KB, x2 on gfx950 for 16-B/lane streams: MI355X_MICROARCH.md) of the sweep contraction per launch for
# several tile maps; one counter per pass, --kernel-trace only.  usage: tools/pmc_sweep_traffic.sh "<tilemap:persist> ..."
export TMPDIR=/tmp
for cfg in ${1:-"3:0"}; do
  tm=${cfg%%:*}; ps=${cfg##*:}
  out=gpurun_out/${TRAFFIC_TAG:-r02}_traffic_${tm}_${ps}_alt${GPRY_SWEEP_ALTWALK:-0}
  mkdir -p $out
  GPRY_SWEEP_PERSIST=$ps timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o t -- python3 tools/prof_sweep.py 4096 16 131072 32768 $tm 3 0 > $out/out.log 2>&1
  python3 - "$out" "$cfg" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print(sys.argv[2], "no counters"); sys.exit(0)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "sweep_gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"tilemap:persist {sys.argv[2]}: {len(v)} launches, FETCH_SIZE per launch {sum(v) / max(len(v), 1) / 1e6:.3f} e6 KB -> x2 = {2 * sum(v) / max(len(v), 1) * 1024 / 1e9:.2f} GB")
PY
done
