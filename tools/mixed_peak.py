#!/usr/bin/env python3
"""FP64 on the matrix pipe and on the vector ALU at the same time (gpry_microbench kind 7)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

dev = _lib.Device(0)
for wpc in (1, 2):
    for mode in (1, 2, 3, 1, 3):
        v = dev.microbench(7, mode + 16 * wpc)
        print(f"wg/CU {wpc} mode {mode}: {v:.2f} TFLOP/s", flush=True)
