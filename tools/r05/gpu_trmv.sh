#!/bin/bash
# round 5: trmv_lower_kernel with eight loads in flight -- per-kernel times of a single evaluation (N = 4096) and of a batch (32 thetas, N = 1024)
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
show() { python3 - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("trmv", "colsum", "lml_traces", "logdet")):
        print(f"{r['Name'][:36]:36s} calls={r['Calls']:>4s} avg={float(r['AverageNs']) / 1e3:7.1f} us")
PY
}
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o p -- python3 $R/tools/prof_factor.py 4096 16 3 > /dev/null 2>&1
echo "single, N = 4096" | tee $R/gpurun_out/r05/trmv_stats.log; show $(find /tmp/pt -name "*kernel_stats.csv" | head -1) | tee -a $R/gpurun_out/r05/trmv_stats.log
rm -rf /tmp/pt2; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt2 -o p -- python3 $R/tools/r04/prof_lml_batch.py 1024 8 32 5 > /dev/null 2>&1
echo "32 thetas, N = 1024" | tee -a $R/gpurun_out/r05/trmv_stats.log; show $(find /tmp/pt2 -name "*kernel_stats.csv" | head -1) | tee -a $R/gpurun_out/r05/trmv_stats.log
