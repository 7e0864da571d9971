#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03_fuzz; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python3 - > $O/fuzz.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "tests/tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import fuzz_parity, fuzz_mirror, fuzz_gates
t0 = time.time()
tot = 0
for seed in range(int(__import__("os").environ.get("FUZZ_SEED0", "100")), int(__import__("os").environ.get("FUZZ_SEED0", "100")) + 12):
    bad, worst = fuzz_parity.run(n_cases=50, seed=seed)
    tot += bad
    print("fuzz_parity seed", seed, "violations", bad, {k: float("%.2e" % v) for k, v in worst.items()}, flush=True)
bad, worst = fuzz_mirror.run(n_seq=80, seed=7)
tot += bad
print("fuzz_mirror violations", bad, worst, flush=True)
bad, n_inf = fuzz_gates.run(n_cases=60, seed=9)
tot += bad
print("fuzz_gates violations", bad, n_inf, flush=True)
print("TOTAL violations", tot, "in %.0f s" % (time.time() - t0))
PY
tail -20 $O/fuzz.log
