"""Seeded synthetic inputs for the node-AVI path.

Every node draws from its own counter-based stream Philox(key=(seed, node_id)), so any shard can
regenerate exactly its nodes.  Math layout here ((rows, cols) numpy); use engine.colmajor() for
the C-ABI."""
from __future__ import annotations

import numpy as np

import qpn_amd  # noqa: F401  (import shim)

SEED = 20240422
INF = np.inf


from qpn_amd.synthetic import node_rng, shared_params, synth_node, synth_nodes  # noqa: E402,F401


def reduced_blocks(Q, R, qd, A, B, l, u, w):
    """numpy statement of the reduced single-node KKT blocks (src/avi.jl:205-251 + :305-377 without
    the dead xi block / slack block): M=[[Q,-A'],[A,0]], q=[qd+R w; B w], kind=[STD n; GAVI m]."""
    count, n, _ = Q.shape
    m = A.shape[1]
    N = n + m
    M = np.zeros((count, N, N))
    M[:, :n, :n] = Q
    M[:, :n, n:] = -np.swapaxes(A, 1, 2)
    M[:, n:, :n] = A
    w = np.asarray(w, dtype=np.float64)
    if w.ndim == 1:
        q = np.concatenate([qd + R @ w, B @ w], axis=1)
    else:                                   # one parameter vector per node
        q = np.concatenate([qd + np.einsum("bnp,bp->bn", R, w), np.einsum("bmp,bp->bm", B, w)], axis=1)
    lo = np.concatenate([np.full((count, n), -INF), l], axis=1)
    hi = np.concatenate([np.full((count, n), INF), u], axis=1)
    kind = np.concatenate([np.zeros((count, n), np.uint8), np.ones((count, m), np.uint8)], axis=1)
    return M, q, lo, hi, kind


def random_box_mcp(rng, N, skew=0.5, p_inf=0.3, p_fix=0.1):
    """Strongly monotone box-MCP (all STD rows) with mixed bound kinds."""
    G = rng.standard_normal((N, N))
    S = rng.standard_normal((N, N))
    M = G @ G.T / N + 0.2 * np.eye(N) + skew * (S - S.T)
    q = 2 * rng.standard_normal(N)
    l = np.where(rng.random(N) < p_inf, -INF, -np.abs(rng.standard_normal(N)))
    u = np.where(rng.random(N) < p_inf, INF, np.abs(rng.standard_normal(N)))
    fx = rng.random(N) < p_fix
    l = np.where(fx, 0.3, l)
    u = np.where(fx, 0.3, u)
    z0 = rng.standard_normal(N)
    return M, q, l, u, z0


def four_player_game(rng):
    """Config 3 structure (examples/four_player_matrix_game.jl:6-176): 8 vars, 4 players with 2
    decision variables each, box +-5 (:123-126), cost_i = sum_j ||offset_ij||^2 (:149-157).
    Returns per-player (Q_i (8x8), q_i (8,)) and the constellation draws."""
    c = rng.standard_normal((4, 4, 2))
    Qs, qs = [], []
    for i in range(4):
        Q = np.zeros((8, 8)); q = np.zeros(8)
        for j in range(4):
            if j == i:
                # d = x_i - c_ii
                for a in range(2):
                    Q[2 * i + a, 2 * i + a] += 2.0
                    q[2 * i + a] += -2.0 * c[i, i, a]
            else:
                # d = x_j - x_i - c_ij
                for a in range(2):
                    ii, jj = 2 * i + a, 2 * j + a
                    Q[ii, ii] += 2.0; Q[jj, jj] += 2.0
                    Q[ii, jj] += -2.0; Q[jj, ii] += -2.0
                    q[jj] += -2.0 * c[i, j, a]
                    q[ii] += 2.0 * c[i, j, a]
        Qs.append(Q); qs.append(q)
    return Qs, qs, c
