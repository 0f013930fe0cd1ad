"""setup(:name) -- the reference's example problems restated as numeric data (the Symbolics
front-end of src/programs.jl:147-201 is out of scope), plus the synthetic nets of SURVEY.md 8(d).

  setup("simple_bilevel")            examples/simple_bilevel.jl:6-35      (BASELINE config 1)
  setup("robust_avoid_simple")       examples/robust_avoid_simple.jl:1-93 (config 2; build-seeded polygons)
  setup("four_player_matrix_game")   examples/four_player_matrix_game.jl:6-176 (config 3)
  setup("synthetic_pairs")           leader-follower pairs, config 4 structure at small size

Julia's MersenneTwister + randn streams cannot be reproduced without Julia, so the random examples
draw from numpy Philox with a recorded seed.
"""
from __future__ import annotations

import numpy as np

from .programs import QPNet

INF = np.inf


def setup(name, **kwargs):
    name = str(name).lstrip(":")
    fn = {"simple_bilevel": _simple_bilevel, "four_player_matrix_game": _four_player_matrix_game,
          "robust_avoid_simple": _robust_avoid_simple, "synthetic_pairs": _synthetic_pairs}.get(name)
    if fn is None:
        raise KeyError(f"unknown example {name}")
    return fn(**kwargs)


def _quad_from_terms(nv, terms):
    """cost = sum_k ||D_k x - c_k||^2  ->  (Q, q, k) with f = 1/2 x'Qx + q'x + k."""
    Q = np.zeros((nv, nv)); q = np.zeros(nv); k = 0.0
    for D, c in terms:
        D = np.atleast_2d(D); c = np.atleast_1d(c)
        Q += 2 * D.T @ D; q += -2 * D.T @ c; k += float(c @ c)
    return Q, q, k


def _simple_bilevel(**kwargs):
    """variables [w1, w2, x, y]; f1 = (y-x)^2 s.t. y >= 0 over y; f2 = ||[x;y]-w||^2 over x; edge 2 -> 1."""
    net = QPNet(4)
    cid = net.add_constraint(np.array([[0, 0, 0, 1.0]]), [0.0], [INF])
    Q1, q1, k1 = _quad_from_terms(4, [(np.array([[0, 0, -1.0, 1.0]]), [0.0])])
    p1 = net.add_qp(Q1, q1, [cid], [3], k1)
    D = np.array([[-1.0, 0, 1, 0], [0, -1.0, 0, 1]])
    Q2, q2, k2 = _quad_from_terms(4, [(D, [0.0, 0.0])])
    p2 = net.add_qp(Q2, q2, [], [2], k2)
    net.add_edges([(p2, p1)])
    net.assign_constraint_groups()
    net.set_options(debug_visualize=False, **kwargs)
    net.default_initialization = np.zeros(4)
    return net


def _four_player_matrix_game(edge_list=(), seed=2, constellations=None, **kwargs):
    """8 variables (x1..x4 in R^2), box +-5 per player (:123-126), cost_i = sum_j ||offset_ij||^2
    (:149-157).  `constellations` (4,4,2) may be given; else drawn from Philox(seed)."""
    if constellations is None:
        constellations = np.random.Generator(np.random.Philox(key=[seed, 7])).standard_normal((4, 4, 2))
    c = np.asarray(constellations, dtype=np.float64)
    net = QPNet(8)
    sel = lambda i: np.eye(8)[2 * i:2 * i + 2]
    for i in range(4):
        cid = net.add_constraint(sel(i), [-5.0, -5.0], [5.0, 5.0])
        terms = []
        for j in range(4):
            if j == i:
                terms.append((sel(i), c[i, i]))
            else:
                terms.append((sel(j) - sel(i), c[i, j]))
        Q, q, k = _quad_from_terms(8, terms)
        net.add_qp(Q, q, [cid], [2 * i, 2 * i + 1], k)
    net.add_edges(list(edge_list))
    net.assign_constraint_groups()
    net.set_options(**kwargs)
    net.default_initialization = np.zeros(8)
    net.problem_data["constellations"] = c
    return net


def _robust_avoid_simple(num_obj=2, num_poly_faces=5, seed=1, max_ego_delta=15.0, max_obj_delta=1.0, **kwargs):
    """18 variables, 5 nodes, 3 levels: ego -> adversary_i -> separating-hyperplane_i.
    Variable order as QPNet(xe, xo, ue, uo, s, eps) at examples/robust_avoid_simple.jl:38."""
    g = np.random.Generator(np.random.Philox(key=[seed, 11]))
    F = num_poly_faces

    def polygon():
        ang = 2 * np.pi * np.arange(F) / F + 0.15 * g.standard_normal(F) + np.pi * g.random()
        return np.stack([np.cos(ang), np.sin(ang)], axis=1), 0.2 + 0.8 * g.random() * np.ones(F)

    Ae, be = polygon()
    polys = [polygon() for _ in range(num_obj)]
    nv = 2 + 2 * num_obj + 2 + 2 * num_obj + 2 * num_obj + num_obj
    ix = {}
    o = 0
    ix["xe"] = [o, o + 1]; o += 2
    ix["xo"] = [[o + 2 * i, o + 2 * i + 1] for i in range(num_obj)]; o += 2 * num_obj
    ix["ue"] = [o, o + 1]; o += 2
    ix["uo"] = [[o + 2 * i, o + 2 * i + 1] for i in range(num_obj)]; o += 2 * num_obj
    ix["s"] = [[o + 2 * i, o + 2 * i + 1] for i in range(num_obj)]; o += 2 * num_obj
    ix["eps"] = [o + i for i in range(num_obj)]; o += num_obj
    net = QPNet(nv)
    s_players, a_players = {}, {}
    for i in range(num_obj):
        Ao, bo = polys[i]
        rows = np.zeros((2 * F, nv)); lb = np.zeros(2 * F)
        # Ae (s_i - (xe+ue)) + be + eps_i >= 0
        rows[:F, ix["s"][i]] = Ae; rows[:F, ix["xe"]] = -Ae; rows[:F, ix["ue"]] = -Ae; rows[:F, ix["eps"][i]] = 1.0
        lb[:F] = -be
        rows[F:, ix["s"][i]] = Ao; rows[F:, ix["xo"][i]] = -Ao; rows[F:, ix["uo"][i]] = -Ao; rows[F:, ix["eps"][i]] = 1.0
        lb[F:] = -bo
        cid = net.add_constraint(rows, lb, np.full(2 * F, INF))
        q = np.zeros(nv); q[ix["eps"][i]] = 1.0                       # cost = eps_i  (LP-like: Q = 0)
        s_players[i] = net.add_qp(np.zeros((nv, nv)), q, [cid], ix["s"][i] + [ix["eps"][i]])
    for i in range(num_obj):
        rows = np.zeros((2, nv)); rows[0, ix["uo"][i][0]] = 1; rows[1, ix["uo"][i][1]] = 1
        cid = net.add_constraint(rows, [-max_obj_delta] * 2, [max_obj_delta] * 2)
        q = np.zeros(nv); q[ix["eps"][i]] = 1.0
        a_players[i] = net.add_qp(np.zeros((nv, nv)), q, [cid], ix["uo"][i])
    rows = np.zeros((2 + num_obj, nv)); rows[0, ix["ue"][0]] = 1; rows[1, ix["ue"][1]] = 1
    for i in range(num_obj):
        rows[2 + i, ix["eps"][i]] = 1
    cid = net.add_constraint(rows, [-max_ego_delta] * 2 + [0.0] * num_obj, [max_ego_delta] * 2 + [INF] * num_obj)
    Qm = np.array([[0.0, 0], [0, 0.001]]); qv = np.array([-1.0, 0.0])
    E = np.zeros((2, nv)); E[0, ix["xe"][0]] = 1; E[0, ix["ue"][0]] = 1; E[1, ix["xe"][1]] = 1; E[1, ix["ue"][1]] = 1
    Q = E.T @ Qm @ E; q = E.T @ qv
    ego = net.add_qp(Q, q, [cid], ix["ue"])
    edges = [(ego, a_players[i]) for i in range(num_obj)] + [(a_players[i], s_players[i]) for i in range(num_obj)]
    net.add_edges(edges)
    net.assign_constraint_groups()
    net.set_options(debug_visualize=False, **kwargs)
    init = np.zeros(nv); init[ix["xe"]] = [-5.0, 0.0]
    for i in range(num_obj):
        init[ix["xo"][i]] = [3.0 * i, -1.0]
    net.default_initialization = init
    net.problem_data.update(Ae=Ae, be=be, polys=polys, index=ix)
    return net


def _synthetic_pairs(pairs=4, n=4, m=4, seed=20240422, first=0, **kwargs):
    """`pairs` independent leader-follower pairs in one two-level net (config 4's structure,
    SURVEY.md section 8(d)): leader k owns n variables and has box rows; follower k owns n variables,
    is coupled to its leader's variables through its cost, and has m two-sided rows.  Costs and rows are given in
    LOCAL form (the 2n variables of the pair), so the net is as large as one likes: 5 000 pairs x (32, 32) are
    10 000 nodes over 320 000 variables.  Pair k draws from Philox(seed, 1000 + first + k): a net of one pair with
    first = k is pair k of the large net on its own."""
    nv = 2 * n * pairs
    net = QPNet(nv)
    leaders, followers = [], []
    for k in range(pairs):
        g = np.random.Generator(np.random.Philox(key=[seed, 1000 + first + k]))
        lv = list(range(2 * n * k, 2 * n * k + n)); fv = list(range(2 * n * k + n, 2 * n * (k + 1)))
        pv = lv + fv                                             # the pair's variables: local positions 0..n-1 leader, n..2n-1 follower
        L, F = slice(0, n), slice(n, 2 * n)
        G = g.standard_normal((n, n)); Qf = G.T @ G / n + 0.1 * np.eye(n)
        Rf = 0.3 * g.standard_normal((n, n))
        Q = np.zeros((2 * n, 2 * n)); Q[F, F] = Qf; Q[F, L] = Rf; Q[L, F] = Rf.T
        q = np.zeros(2 * n); q[F] = g.standard_normal(n)
        A = g.standard_normal((m, n)) / np.sqrt(n)
        cid = net.add_constraint(A, -1 - np.abs(g.standard_normal(m)), 1 + np.abs(g.standard_normal(m)), cols=fv)
        followers.append(net.add_qp(Q, q, [cid], fv, idx=pv))
        G2 = g.standard_normal((n, n)); Ql = G2.T @ G2 / n + 0.5 * np.eye(n)
        Q2 = np.zeros((2 * n, 2 * n)); Q2[L, L] = Ql; Q2[F, F] = 0.1 * np.eye(n)
        q2 = np.zeros(2 * n); q2[L] = g.standard_normal(n)
        cid2 = net.add_constraint(np.eye(n), -2 * np.ones(n), 2 * np.ones(n), cols=lv)
        leaders.append(net.add_qp(Q2, q2, [cid2], lv, idx=pv))
    net.add_edges([(leaders[k], followers[k]) for k in range(pairs)])
    net.assign_constraint_groups()
    net.set_options(**kwargs)
    return net
