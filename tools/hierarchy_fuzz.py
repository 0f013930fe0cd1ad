#!/usr/bin/env python3
"""Developer aid: the reference's four_player_matrix_game example under RANDOM hierarchies (random DAG edge lists over the four players)
and random payoff seeds, through the whole host loop -- against the closed form of tests/test_analytic_equilibria.py
(backward substitution: valid while the example's box stays inactive).  SCALE=s multiplies the payoff constellations, so that
equilibria leave the +-5 box and constraints become active on every level: those draws are compared with the same net solved on
the CPU oracle engine (same algorithm, deterministic: the same equilibrium is expected).  ENGINE=oracle runs the CPU engine alone
(no GPU needed).  Usage: python tools/hierarchy_fuzz.py [trials] [seed]"""
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
scale = float(os.environ.get("SCALE", "1"))
ref_eng = None
if scale != 1.0 and os.environ.get("ENGINE") != "oracle":
    from oracle_engine import OracleEngine
    ref_eng = OracleEngine()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
pairs = [(i, j) for i, j in itertools.combinations(range(1, 5), 2)]          # i < j: every subset is a DAG
worst = 0.0; closed = 0; other = 0; unsolved = 0; bad = []
for t in range(trials):
    edges = [e for e in pairs if rng.random() < 0.35]
    seed = int(rng.integers(1, 10_000))
    try:
        cons = scale * np.random.Generator(np.random.Philox(key=[seed, 7])).standard_normal((4, 4, 2))
        net = examples.setup("four_player_matrix_game", edge_list=edges, seed=seed, constellations=cons)
        ret = algorithm.solve(net, engine=eng)
        if not ret["solved"]:
            # the reference's own outer loop gives up on some nets ("Cycling detected", src/algorithm.jl): a failure only if the CPU
            # engine does not end the same way
            if ref_eng is not None:
                net2 = examples.setup("four_player_matrix_game", edge_list=edges, seed=seed, constellations=cons)
                ref = algorithm.solve(net2, engine=ref_eng)
                if not ref["solved"] and ref.get("error") == ret.get("error"):
                    unsolved += 1; continue
            bad.append((edges, seed, f"not solved: {ret.get('error')}")); continue
        want = backward_substitution(net)
        if np.max(np.abs(want)) < 5.0 - 1e-6:
            worst_before = worst
            d = float(np.max(np.abs(ret["x_opt"] - want))); worst = max(worst, d); closed += 1
            if d > 1e-8:
                # with an enlarged payoff a net can have several equilibria (the closed form is the one with every box inactive; the
                # loop, started from 0, may settle where a follower's box is active): then the CPU engine decides
                ok2 = False
                if ref_eng is not None:
                    net2 = examples.setup("four_player_matrix_game", edge_list=edges, seed=seed, constellations=cons)
                    ref = algorithm.solve(net2, engine=ref_eng)
                    ok2 = ref["solved"] and float(np.max(np.abs(ret["x_opt"] - ref["x_opt"]))) <= 1e-8
                if ok2:
                    worst = worst_before; closed -= 1; other += 1
                else:
                    bad.append((edges, seed, f"differs from the closed form by {d:.2e}"))
        else:
            other += 1
            if ref_eng is not None:
                net2 = examples.setup("four_player_matrix_game", edge_list=edges, seed=seed, constellations=cons)
                ref = algorithm.solve(net2, engine=ref_eng)
                d = float(np.max(np.abs(ret["x_opt"] - ref["x_opt"])))
                if not ref["solved"] or d > 1e-8:
                    bad.append((edges, seed, f"oracle engine: solved {ref['solved']}, difference {d:.2e}"))
    except Exception as e:
        bad.append((edges, seed, f"{type(e).__name__}: {str(e)[:120]}"))
for b in bad:
    print("  ", b)
print(f"{trials} random hierarchies: {closed} against the closed form (worst {worst:.2e}), {other} with an active box (against the oracle engine where both are here), {unsolved} given up by both engines with the same message, {len(bad)} failures")
