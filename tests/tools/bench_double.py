"""Child script of tests/test_bench_launcher_cpu.py: ``bench.py`` with the oracle-backed device double
(tests/oracle_device.py) in place of ``gpry_amd._lib.Device``, so that the launcher of ``bench.py --gpus N``
(environment of the ranks, relay of rank 0's line, exit codes, watchdog) runs end to end without a GPU.
Test infrastructure only; ``GPRY_BENCH_DOUBLE_FAIL_RANK`` makes that rank exit 7 before anything else,
``GPRY_BENCH_DOUBLE_FAIL_MID_RANK`` / ``_FAIL_LATE_RANK`` in its first step / behind the timed region (exit 8 / 9)."""
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

# a rank that dies AFTER the rendezvous (first step) or AFTER the timed region (behind the closing all-reduce, which it
# still completes so that its peers get past it): what the launcher's error object is for
_me = os.environ.get("RANK", "0")
if os.environ.get("GPRY_BENCH_DOUBLE_FAIL_MID_RANK", "") == _me:
    def _die_mid(*a, **k):
        print("bench_double: this rank fails in its first step", file=sys.stderr, flush=True)
        os._exit(8)
    bench.rewind = _die_mid
if os.environ.get("GPRY_BENCH_DOUBLE_FAIL_LATE_RANK", "") == _me:
    _real = bench._GlooComm.allreduce_max

    def _die_late(self, arr):
        out = _real(self, arr)
        print("bench_double: this rank fails behind the timed region", file=sys.stderr, flush=True)
        os._exit(9)
    bench._GlooComm.allreduce_max = _die_late

bench.main(sys.argv[1:])
