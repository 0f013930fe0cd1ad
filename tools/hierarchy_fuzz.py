#!/usr/bin/env python3
"""Developer aid: the reference's four_player_matrix_game example under RANDOM hierarchies (random DAG edge lists over the four players)
and random payoff seeds, through the whole host loop -- against the closed form of tests/test_analytic_equilibria.py
(backward substitution: valid while the example's box stays inactive; other draws are only checked for `solved` and the fixed
point).  ENGINE=oracle runs the CPU engine (no GPU needed).  Usage: python tools/hierarchy_fuzz.py [trials] [seed]"""
import itertools, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
from qpn_amd import algorithm, examples
from test_analytic_equilibria import backward_substitution
if os.environ.get("ENGINE") == "oracle":
    from oracle_engine import OracleEngine
    eng = OracleEngine()
else:
    eng = qpn_amd.default_engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
pairs = [(i, j) for i, j in itertools.combinations(range(1, 5), 2)]          # i < j: every subset is a DAG
worst = 0.0; closed = 0; other = 0; bad = []
for t in range(trials):
    edges = [e for e in pairs if rng.random() < 0.35]
    seed = int(rng.integers(1, 10_000))
    try:
        net = examples.setup("four_player_matrix_game", edge_list=edges, seed=seed)
        ret = algorithm.solve(net, engine=eng)
        if not ret["solved"]:
            bad.append((edges, seed, "not solved")); continue
        want = backward_substitution(net)
        if np.max(np.abs(want)) < 5.0 - 1e-6:
            d = float(np.max(np.abs(ret["x_opt"] - want))); worst = max(worst, d); closed += 1
            if d > 1e-8:
                bad.append((edges, seed, f"differs from the closed form by {d:.2e}"))
        else:
            other += 1
    except Exception as e:
        bad.append((edges, seed, f"{type(e).__name__}: {str(e)[:120]}"))
for b in bad:
    print("  ", b)
print(f"{trials} random hierarchies: {closed} against the closed form (worst {worst:.2e}), {other} with an active box (solved only), {len(bad)} failures")
