#!/usr/bin/env python3
"""Writes the committed fixtures under tests/golden/.

 * simple_bilevel_cases.json -- DATA of the reference's own end-to-end test,
   /root/reference/test/simple_bilevel.jl:4-16 (parameter vectors w, accepted equilibria (x,y),
   start point): inputs and expected outputs only, no reference source text.
 * avi_kats.json -- hand-derived AVI known answers of SURVEY.md section 8(c)(2): config-1 level-2
   AVI in the reference form z = [y, xi, lam, s] (src/avi.jl:113-128, :244, :356-367).
 * oracle_vectors.json -- seeded box-MCP / GAVI instances with the CPU oracle's answers.  The
   reference (Julia + PATH) cannot run here, so these are NOT reference outputs: they freeze the
   oracle's behaviour (regression) and are cross-checked against independent solvers in
   tests/test_oracle_crosscheck.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding as ob  # noqa: E402
import problems as P  # noqa: E402


def enc(a):
    a = np.asarray(a, dtype=np.float64)
    return [[("inf" if v == np.inf else "-inf" if v == -np.inf else float(v)) for v in row] for row in np.atleast_2d(a)]


def main():
    r2 = float(np.sqrt(2.0))
    cases = {
        "source": "test/simple_bilevel.jl:4-16 (data only)",
        "variables": ["w1", "w2", "x", "y"], "x0": [0.0, 0.0], "atol": 1e-4,
        "w": [[-2, -3], [0, -1], [1, -3], [1, -1], [1, 0], [0, 1], [-1, 1 + r2], [0, 0]],
        "accepted_xy": [[[-2, 0]], [[0, 0]], [[0, 0]], [[0, 0]], [[.5, .5]], [[.5, .5], [0, 0]],
                        [[-1, 0], [r2 / 2, r2 / 2]], [[0, 0]]],
        "min_pieces_root_graph": [1, 2, 1, 2, 1, 1, 1, 3],
    }
    json.dump(cases, open(os.path.join(HERE, "simple_bilevel_cases.json"), "w"), indent=1)

    kats = {"source": "SURVEY.md section 8(c)(2), hand-derived",
            "M": [[0, 1, 0, 0], [2, 0, -1, 0], [1, 0, 0, -1], [0, 0, 1, 0]],
            "l": ["-inf", "-inf", "-inf", 0.0], "u": ["inf"] * 4,
            "cases": [{"x": -2.0, "q": [0, 4.0, 0, 0], "z": [0, 0, 4, 0], "slack_row_codes": [1]},
                      {"x": 0.5, "q": [0, -1.0, 0, 0], "z": [.5, 0, 0, .5], "slack_row_codes": [2]},
                      {"x": 0.0, "q": [0, 0.0, 0, 0], "z": [0, 0, 0, 0], "slack_row_codes": [1, 2]}]}
    json.dump(kats, open(os.path.join(HERE, "avi_kats.json"), "w"), indent=1)

    rng = np.random.default_rng(20240422)
    vecs = []
    for t in range(24):
        if t % 2 == 0:
            N = int(rng.integers(2, 13))
            M, q, l, u, z0 = P.random_box_mcp(rng, N)
            kind = np.zeros(N, np.uint8)
        else:
            n, m = int(rng.integers(1, 6)), int(rng.integers(1, 8))
            Q, R, qd, A, B, lo, hi = P.synth_node(5000 + t, n, m)
            Mb, qb, lb, ub, kb = P.reduced_blocks(Q[None], R[None], qd[None], A[None], B[None], lo[None], hi[None], P.shared_params())
            M, q, l, u, kind = Mb[0], qb[0], lb[0], ub[0], kb[0]
            z0 = np.zeros(n + m)
        r = ob.solve_avi(M, q, l, u, z0=z0, kind=kind)
        vecs.append(dict(M=enc(M), q=enc(q)[0], l=enc(l)[0], u=enc(u)[0], z0=enc(z0)[0], kind=[int(k) for k in kind],
                         z=enc(r["z"])[0], status=r["status"], active=[int(a) for a in r["active"]], pivots=r["pivots"]))
    json.dump({"source": "CPU oracle (NOT reference outputs; regression + cross-checked)", "vectors": vecs},
              open(os.path.join(HERE, "oracle_vectors.json"), "w"))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
