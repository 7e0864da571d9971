#!/bin/bash
# what the driver runs at the end of the round: smoke, the -m gpu suite (tools/r06/gpu_suite.sh), the bench with its default arguments
mkdir -p gpurun_out/r06
{
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
[ "$1" = "suite" ] && timeout 3000 python3 -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
t0=$(date +%s)
timeout 1200 python3 bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
echo "bench.py with default arguments: $(( $(date +%s) - t0 )) s wall"
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r06/bench_default.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "vs_baseline")})
print("roofline", {k: r["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "traffic")}, "cpu_baseline", {k: r["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")})
print("cycle", r["cycle"]["panel_form"], r["cycle"]["runner_cadence_ms"], r["refit_extras"]["full_fit_default_restarts"]["ms"], r["refit_extras"]["full_fit_default_restarts"].get("ms_warm"))
PY
tail -5 gpurun_out/r06/bench_default.err | cut -c1-300
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/final.log
