"""Randomised parity sweep (tests/tools/fuzz_parity.py): random N (around the 64/128 padding quanta), d in
1..32, M across the small-batch / panel path boundaries, all four kernels, masks, chunk sizes."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_against_the_oracle(seed):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_parity
    bad, worst = fuzz_parity.run(n_cases=25, seed=seed)
    assert bad == 0, worst
    assert worst["mean"] < 1e-8 and worst["var"] < 1e-9


def test_random_operation_sequences_on_the_host_mirror():
    """tests/tools/fuzz_mirror.py: appends, predictions, conditioned models, copies / pickles and NORA
    proposals of the mirror classes on the GPU against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_mirror
    bad, worst = fuzz_mirror.run(n_seq=12, seed=42)
    assert bad == 0, worst


def test_random_gated_models_device_gates_equal_host_masks():
    """tests/tools/fuzz_gates.py: random -inf half-spaces (SVM) and trust regions; NORA with the gates on
    the device equals NORA with libsvm + numpy masks."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_gates
    bad, n_inf = fuzz_gates.run(n_cases=20, seed=5)
    assert bad == 0 and n_inf > 1000


def test_factorisation_stress_from_three_threads():
    """tests/tools/stress_factor.py: random sizes / kernels / dimensions factorised and evaluated from three
    host threads (three device contexts) at once; every factor checked (L L^T = K, V L = I).  Exercises the
    counter- and flag-synchronised panel kernel under contention (60 s runs: 1447 models, no failure)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "stress_factor.py"), "6", "3"],
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "failures: 0" in out.stdout, out.stdout
