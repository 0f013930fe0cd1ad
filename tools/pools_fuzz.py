#!/usr/bin/env python3
"""Developer aid: qpn_assemble_pools (combine_gavis, src/avi.jl:305-377) on random pools with shared variables and batches -- the
suite's own random-pool check (tests/test_gpu_pools.py) over many more seeds.  Usage: python tools/pools_fuzz.py [seeds] [first]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import qpn_amd
import test_gpu_pools as T
eng = qpn_amd.default_engine(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for s in range(first, first + seeds):
    try:
        T.test_random_pools_with_shared_variables_and_batches.__wrapped__(eng, s) if hasattr(T.test_random_pools_with_shared_variables_and_batches, "__wrapped__") else T.test_random_pools_with_shared_variables_and_batches(eng, s)
    except AssertionError as e:
        bad += 1; print("  seed", s, "FAILED:", str(e)[:200], flush=True)
print(f"{seeds} seeds of random pools: {bad} failures")
