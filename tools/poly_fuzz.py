#!/usr/bin/env python3
"""Developer aid: the polyhedral LP primitives (row F3: exemplar / isempty / slack rule / implicit bounds) on the HIP engine
over many seeds of the test suite's own random polyhedra, against HiGHS (the checks of tests/test_polyhedra.py).
Usage: python tools/poly_fuzz.py [seeds] [first]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import qpn_amd
import test_polyhedra as T
eng = qpn_amd.default_engine(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = []
for s in range(first, first + seeds):
    for name, fn in (("min-norm", T._check), ("slack+implicit", T._check_slack_and_implicit)):
        try:
            fn(eng, s)
        except AssertionError as e:
            bad.append((s, name, str(e)[:200]))
            print("  seed", s, name, "FAILED:", str(e)[:200], flush=True)
print(f"{seeds} seeds x 2 checks: {len(bad)} failures")
