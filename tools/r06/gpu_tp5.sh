#!/bin/bash
mkdir -p gpurun_out/r06
{
timeout 1200 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fitted_model_of_the_bench or panel_form" 2>&1 | tail -15
timeout 1500 python3 -m pytest tests/test_lml_batch_gpu.py -x -q -m gpu -k "throughput" 2>&1 | tail -5
GPRY_HIP_FIT_SCHEDULE=throughput timeout 600 python3 tools/r06/time_fit.py 4096 16 2
GPRY_HIP_FIT_SCHEDULE=throughput GPRY_HIP_FIT_TP_GROUPS=2 timeout 600 python3 tools/r06/time_fit.py 4096 16 2
GPRY_HIP_FIT_SCHEDULE=latency timeout 600 python3 tools/r06/time_fit.py 4096 16 2
timeout 900 python3 bench.py --steps 10 --warmup 3 > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err; tail -3 gpurun_out/r06/bench_a.err
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r06/bench_a.json").read().strip().splitlines()[-1])
c = r["cycle"]
print("ms_per_step", r["ms_per_step"], "roofline", r["roofline"]["frac"], "panel", c.get("panel_form"), c.get("panel_error_estimate"), c.get("panel_error_variance"))
print("refit", c["refit_ms"], "acq", c["acquisition_ms"], "one_lml", c["one_lml_grad_call_ms"], "cadence", c.get("runner_cadence_ms"), c.get("runner_cadence"))
print("stages", c["stage_ms_per_step"])
print("refit_extras", json.dumps(r.get("refit_extras"))[:1500])
print("cholesky", json.dumps(r.get("cholesky"))[:800])
sn = r.get("small_n", {})
for k, v in sn.items():
    if isinstance(v, dict) and "fit_full_ms" in v: print(k, {q: v[q] for q in v if q.startswith("fit_full") or "batch32" in q or q.startswith("lml_grad")})
PY
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp5.log
