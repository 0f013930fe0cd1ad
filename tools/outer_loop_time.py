#!/usr/bin/env python3
"""Developer aid: wall time of algorithm.solve on synthetic_pairs nets on the HIP engine (no profiler), the share inside the C-ABI.
usage: python tools/outer_loop_time.py [pairs n m]..."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import qpn_amd
from qpn_amd import algorithm, examples
warnings.simplefilter("ignore")
args = [int(v) for v in sys.argv[1:]] or [1000, 16, 16, 2000, 16, 16, 250, 32, 32]
eng = qpn_amd.default_engine(0)
algorithm.solve(examples.setup("synthetic_pairs", pairs=8, n=8, m=8), engine=eng)      # (warm: module loads, first launches)
for i in range(0, len(args), 3):
    pairs, n, m = args[i:i + 3]
    net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
    eng.calls.clear(); eng.seconds.clear()
    t = time.time()
    ret = algorithm.solve(net, engine=eng)
    dt = time.time() - t
    abi = sum(eng.seconds.values())
    print(f"pairs={pairs} n={n} m={m}: solved {ret['solved']} in {dt:.2f} s; inside the C-ABI {abi:.2f} s ({100 * abi / dt:.1f} %), "
          f"{sum(eng.calls.values())} calls", flush=True)
