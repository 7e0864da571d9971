"""Child script of tests/test_bench_launcher_cpu.py: ``bench.py`` with the oracle-backed device double
(tests/oracle_device.py) in place of ``gpry_amd._lib.Device``, so that the launcher of ``bench.py --gpus N``
(environment of the ranks, relay of rank 0's line, exit codes, watchdog) runs end to end without a GPU.
Test infrastructure only; ``GPRY_BENCH_DOUBLE_FAIL_RANK`` makes that rank exit 7 before anything else."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if os.environ.get("GPRY_BENCH_DOUBLE_FAIL_RANK", "") == os.environ.get("RANK", "0"):
    print("bench_double: this rank fails on purpose", file=sys.stderr)
    sys.exit(7)
if os.environ.get("GPRY_BENCH_DOUBLE_HANG_RANK", "") == os.environ.get("RANK", "0"):
    time.sleep(600)

from gpry_amd import _lib          # noqa: E402
from oracle_device import OracleDevice      # noqa: E402


class BenchDouble(OracleDevice):
    def timing_reset(self):
        pass

    def timing(self, name):
        return 0.0, 0

    def microbench(self, kind, nbytes=0):
        raise _lib.GpryHipError("no micro-benchmarks on the double")


def _no_rccl():
    raise _lib.GpryHipError("no RCCL on the double")


_lib.Device = BenchDouble
_lib.device_count = lambda: int(os.environ.get("WORLD_SIZE", "1"))
_lib.RcclComm.unique_id = staticmethod(_no_rccl)

import bench                      # noqa: E402

bench.main(sys.argv[1:])
