#!/usr/bin/env python3
"""cProfile of a few bench cycles: where the host time of a cycle goes."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--cpu-baseline", "off"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats("gpry_amd|numpy", 30)
