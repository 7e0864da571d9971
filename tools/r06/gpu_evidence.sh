#!/bin/bash
# round 6: evidence of the build -- GPU suite, bench lines (default / config1 / farm) and the default + config1 commands under
# rocprofv3 (kernel stats), PMC passes (HBM-side traffic of the sweep), stage timers, the batched objective in both schedules
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $R/gpurun_out/r06/all_gpu_tests.log
cat $R/gpurun_out/r06/all_gpu_tests.log
# HBM-side traffic of the sweep contraction of THIS build first: bench.py carries it (profiles/r06_traffic.json)
cd $R; bash tools/r06/pmc_traffic.sh > /dev/null 2>&1
python3 - <<'PY'
import json, re
rows = [l.rstrip("\n").split("\t") for l in open("gpurun_out/r06/pmc_traffic.txt") if l.count("\t") >= 4]
val = {}
for tag, c, k, n, v in rows:
    key = "sweep" if "sweep_gemm" in k else "cross" if "cross_build" in k else None      # (not sweep_finish_kernel)
    if key: val[(tag, c, key)] = (int(n.split()[1]), float(v.split()[2]))
out = {"_comment": "HBM-side (L2 memory-side: Infinity-Cache hits included) traffic of the sweep contraction of the ROUND-6 build: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, program directly behind `--` (tools/r06/pmc_traffic.sh -> profiles/r06_pmc_traffic_raw.tsv).  FETCH_SIZE (KB) is doubled as MI355X_MICROARCH.md (HBM) prescribes for 16-B/lane streaming reads on gfx950; WRITE_SIZE as reported.  bench.py reads this file (measured_traffic)."}
for tag, Np, d, cand in (("config2", 4096, 16, 32768), ("config1", 1024, 8, 100000)):
    f, w = val.get((tag, "FETCH_SIZE", "sweep")), val.get((tag, "WRITE_SIZE", "sweep"))
    if not f or not w: continue
    candp = -(-cand // 128) * 128
    out[tag] = {"Np": Np, "d": d, "candidates_per_launch": cand, "launches_profiled": f[0],
                "fetch_size_kb_raw": f[1], "write_size_kb_raw": w[1],
                "hbm_bytes_per_launch": int(2 * f[1] * 1024 + w[1] * 1024),
                "algorithmic_bytes_per_launch": int(8 * (Np * Np // 2 + Np * candp) + 8 * (Np // 128) * candp),
                "source": "profiles/r06_traffic.json: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, round-6 build (tools/r06/pmc_traffic.sh)"}
json.dump(out, open("profiles/r06_traffic.json", "w"), indent=1)
json.dump(out, open("gpurun_out/r06/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_comment"}))
PY
cd /tmp
timeout 1200 python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r06/bench_final.json 2> $R/gpurun_out/r06/bench_final.err
rm -rf /tmp/prof_b
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o p -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-baseline off --extras off > $R/gpurun_out/r06/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06/bench_kernel_stats.csv
timeout 600 python3 $R/bench.py --workload config1 --steps 20 --warmup 5 > $R/gpurun_out/r06/bench_config1.json 2> $R/gpurun_out/r06/bench_config1.err
rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o p -- python3 $R/bench.py --workload config1 --steps 20 --warmup 5 --cpu-baseline off > $R/gpurun_out/r06/bench_config1_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06/bench_config1_kernel_stats.csv
timeout 900 python3 $R/bench.py --workload farm --steps 1 --warmup 0 > $R/gpurun_out/r06/farm.json 2> $R/gpurun_out/r06/farm.err
cd $R
timeout 600 python3 tools/r04/time_potrf.py 512 1024 2048 3072 4096 5120 6144 7168 8192 2>&1 | tail -1 > gpurun_out/r06/time_potrf.log
timeout 900 python3 tools/r04/time_lml_batch.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/r06/time_lml_batch.log
{ echo "# the same calls in the throughput schedule (GPRY_HIP_OPTIONS=lml_schedule=1: B thetas in ONE gpry_lml_batch | B x gpry_lml of the latency schedule | ratio)";
  GPRY_HIP_OPTIONS="lml_schedule=1" GPRY_HIP_FIT_SCHEDULE=throughput timeout 900 python3 tools/r04/time_lml_batch.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; } >> gpurun_out/r06/time_lml_batch.log
timeout 300 python3 tools/r04/ab_pipeline_now.py 4096 8192 2>&1 | grep "^N=" > gpurun_out/r06/ab_pipeline.log
cat gpurun_out/r06/time_potrf.log gpurun_out/r06/pmc_traffic.txt; tail -30 gpurun_out/r06/time_lml_batch.log; cat gpurun_out/r06/ab_pipeline.log; head -c 400 gpurun_out/r06/farm.json
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r06/bench_final.json").read().strip().splitlines()[-1])
c = r["cycle"]
print("\nms_per_step", r["ms_per_step"], "value", r["value"], "roofline", r["roofline"]["frac"], "panel", c.get("panel_form"), c.get("panel_error_estimate"), c.get("panel_error_variance"))
print("refit", c["refit_ms"], "acq", c["acquisition_ms"], "one_lml", c["one_lml_grad_call_ms"], "cadence", c.get("runner_cadence_ms"), c.get("runner_cadence"))
print("stages", c["stage_ms_per_step"])
print("kernel_build", r["kernel_build"]["frac"], "cholesky", r["cholesky"]["frac"], r["cholesky"].get("serial_chain"))
print("refit_extras", json.dumps(r.get("refit_extras"))[:1800])
for k, v in r.get("small_n", {}).items():
    if isinstance(v, dict) and "lml_grad_us" in v: print(k, {q: (round(v[q], 2) if isinstance(v[q], float) else v[q]) for q in v})
print("cpu_baseline", json.dumps(r.get("cpu_baseline"))[:600])
PY
