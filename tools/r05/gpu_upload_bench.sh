#!/bin/bash
# round 5: pipelined pool upload -- parity tests, then the default bench line (fresh pool) and the same with the upload in front
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_host_mirror_gpu.py -x -q -m gpu -k "uploaded_underneath or device_gates_match or multi_chunk" 2>&1 | tail -3 | tee gpurun_out/r05/upload_tests.log
timeout 600 python3 bench.py --steps 10 --warmup 3 --extras off --cpu-baseline off > gpurun_out/r05/bench_fresh.json 2> gpurun_out/r05/bench_fresh.err
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/r05/bench_fresh.json"))
c = r["cycle"]
print("fresh", c["fresh_pool_ms_per_step"], "resident", c["resident_pool_ms_per_step"], "refit", c["refit_ms"], "acq", c["acquisition_ms"], c["pool_upload"])
PY
GPRY_HIP_OPTIONS="sweep_upload=0" timeout 600 python3 bench.py --steps 10 --warmup 3 --extras off --cpu-baseline off > gpurun_out/r05/bench_fresh_front.json 2> gpurun_out/r05/bench_fresh_front.err
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/r05/bench_fresh_front.json"))
c = r["cycle"]
print("upload in front: fresh", c["fresh_pool_ms_per_step"], "resident", c["resident_pool_ms_per_step"], c["pool_upload"])
PY
