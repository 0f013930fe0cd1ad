#!/usr/bin/env python3
"""Writes tests/golden/robust_avoid_pieces.json: child pieces and an iterate for the pool AVIs of
setup(:robust_avoid_simple) (BASELINE config 2; structure per examples/robust_avoid_simple.jl:1-93 with build-seeded
polygons).

The reference (Julia + PATH + CDD) cannot run here, so these are NOT reference outputs.  A piece here is the polyhedron
on which a node's active-set recipe stays what it is at the current solution -- the kind of set local_piece
(src/avi_solutions.jl:400-496) describes before the duals are projected out: the node's constraint rows (own rows, then
its children's pieces), rows active at the solution as equalities, the others with their bounds.  Generated bottom-up
with the CPU oracle engine: level 3 solved at the default initialisation, its pieces handed to level 2, level 2 solved,
its pieces handed to level 1.  Run from the repo root:  python tests/golden/make_robust_avoid_pieces.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import qpn_amd  # noqa: E402,F401
from qpn_amd import avi, examples  # noqa: E402
from qpn_amd.programs import Poly  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402


def piece_of(net, pid, S, x, tol=1e-7):
    rows = [net.constraints[ci].poly.vectorize() for ci in net.qps[pid].constraint_indices]
    for j in sorted(net.network_edges[pid]):
        rows.append(S[j].vectorize())
    A = np.vstack([r[0] for r in rows]); l = np.concatenate([r[1] for r in rows]); u = np.concatenate([r[2] for r in rows])
    ax = A @ x
    l2, u2 = l.copy(), u.copy()
    for i in range(len(l)):
        if np.isfinite(l[i]) and abs(ax[i] - l[i]) <= tol:
            u2[i] = l[i]
        elif np.isfinite(u[i]) and abs(ax[i] - u[i]) <= tol:
            l2[i] = u[i]
    return Poly(A, l2, u2, normalise=False)


def main():
    eng = OracleEngine()
    net = examples.setup("robust_avoid_simple")
    x = net.default_initialization.copy()
    S = {}
    for level in (3, 2):
        pool = sorted(net.network_depth_map[level])
        x = avi.solve_qep(net, pool, x, S, engine=eng, reference_form=True)
        for pid in pool:
            S[pid] = piece_of(net, pid, S, x)
    enc = lambda v: [("inf" if t == np.inf else "-inf" if t == -np.inf else float(t)) for t in v]
    out = {"source": "build-seeded (not reference output): tests/golden/make_robust_avoid_pieces.py",
           "x": [float(t) for t in x],
           "pieces": {str(pid): {"A": [[float(t) for t in row] for row in S[pid].A], "l": enc(S[pid].l), "u": enc(S[pid].u)}
                      for pid in S}}
    json.dump(out, open(os.path.join(HERE, "robust_avoid_pieces.json"), "w"), indent=0)
    for pid in S:
        print("node", pid, "piece rows", S[pid].A.shape[0], "equalities", int(np.sum(S[pid].l == S[pid].u)))
    # sizes of the pool AVIs with these pieces
    for level in (3, 2, 1):
        pool = sorted(net.network_depth_map[level])
        dec = sorted(set().union(*[set(net.decision_inds(i)) for i in pool]))
        par = [i for i in range(net.num_vars) if i not in set(dec)]
        lab = {i: avi.create_labeled_gavi_from_qp(net, i, S) for i in pool}
        g = avi.combine_gavis(net.num_vars, dec, par, lab)
        z, st, info = avi.solve_gavi(g, np.concatenate([x[dec], np.zeros(g.M.shape[1] - len(dec))]), x[par], engine=eng, reference_form=True)
        print("level", level, "pool", pool, "N_ref", len(g.l1) + 2 * len(g.l2), "status", int(st), "resid", info["resid"])


if __name__ == "__main__":
    main()
