#!/usr/bin/env python3
"""Share of the non-trivial lines of each host module that also occur verbatim in a module of the reference package
(run where /root/reference exists; a development check, not part of any test)."""
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/gpry"
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpry_amd")


def lines(path):
    keep = []
    for raw in open(path, errors="ignore"):
        t = raw.strip()
        if len(t) >= 8 and not t.startswith("#") and t not in ('"""', "'''"):
            keep.append(t)
    return keep


for name in sorted(os.listdir(here)):
    if not name.endswith(".py"):
        continue
    mine = lines(os.path.join(here, name))
    best = (0.0, None)
    for other in os.listdir(ref):
        if other.endswith(".py") and mine:
            theirs = set(lines(os.path.join(ref, other)))
            frac = sum(x in theirs for x in mine) / len(mine)
            if frac > best[0]:
                best = (frac, other)
    print(f"{name:26s} {len(mine):5d} lines, {best[0]:.2f} identical to {best[1]}")
