#!/usr/bin/env python3
"""Developer aid: BASELINE config 2 (setup(:robust_avoid_simple): 18 variables, 5 nodes, 3 levels, LP-like Q = 0 nodes) over
many seeds (other obstacle polygons) and a few shapes (obstacles, polygon faces) through the whole host loop: HIP engine against
the CPU oracle engine -- both solved to the same point (1e-8), or both given up with the same message.
Usage: python tools/robust_avoid_fuzz.py [seeds] [first]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
from qpn_amd import algorithm, examples
from oracle_engine import OracleEngine
hip = qpn_amd.default_engine(0); cpu = OracleEngine()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 4
same = both_up = 0; bad = []; worst = 0.0
for seed in range(first, first + seeds):
    kw = dict(seed=seed)
    if seed % 5 == 0: kw["num_poly_faces"] = 4 + seed % 4
    if seed % 7 == 0: kw["num_obj"] = 1 + seed % 3
    try:
        a = algorithm.solve(examples.setup("robust_avoid_simple", **kw), engine=hip)
        b = algorithm.solve(examples.setup("robust_avoid_simple", **kw), engine=cpu)
    except Exception as e:
        bad.append((kw, f"{type(e).__name__}: {str(e)[:140]}")); continue
    if a["solved"] and b["solved"]:
        d = float(np.max(np.abs(a["x_opt"] - b["x_opt"]))); worst = max(worst, d)
        if d <= 1e-8: same += 1
        else: bad.append((kw, f"points differ by {d:.2e}"))
    elif not a["solved"] and not b["solved"] and a.get("error") == b.get("error"):
        both_up += 1
    else:
        bad.append((kw, f"HIP solved {a['solved']} ({a.get('error')}), CPU solved {b['solved']} ({b.get('error')})"))
for x in bad:
    print("  ", x)
print(f"{seeds} nets: {same} solved to the same point on both engines (worst difference {worst:.2e}), {both_up} given up by both with the same message, {len(bad)} failures")
