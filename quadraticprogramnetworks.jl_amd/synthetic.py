"""Seeded synthetic node records for the node-AVI path.

Every node draws from its own counter-based stream Philox(key=(seed, node_id)), so any shard (rank)
regenerates exactly its own nodes and no M/q traffic ever crosses GPUs.  Arrays are in math layout
((rows, cols)); use engine.colmajor() for the C-ABI.

Per node (fp64):  G ~ N(0,1)^{n x n},  Q = G'G/n + 0.1 I  (SPD => unique solution => well-posed
active-set parity);  R ~ 0.1 N(0,1)^{n x p} couples p = 8 shared level-1 parameters w ~ N(0,1);
q ~ N(0,1)^n;  A ~ N(0,1)^{m x n}/sqrt(n);  l = -1-|N(0,1)|,  u = 1+|N(0,1)|;  B = 0.
"""
from __future__ import annotations

import numpy as np

SEED = 20240422


def node_rng(node_id: int, seed: int = SEED):
    return np.random.Generator(np.random.Philox(key=[seed, node_id]))


def shared_params(p: int = 8, seed: int = SEED):
    return np.random.Generator(np.random.Philox(key=[seed, 2 ** 40])).standard_normal(p)


def synth_node(node_id: int, n: int, m: int, p: int = 8, seed: int = SEED):
    g = node_rng(node_id, seed)
    G = g.standard_normal((n, n))
    Q = G.T @ G / n + 0.1 * np.eye(n)
    R = 0.1 * g.standard_normal((n, p))
    qd = g.standard_normal(n)
    A = g.standard_normal((m, n)) / np.sqrt(n)
    l = -1.0 - np.abs(g.standard_normal(m))
    u = 1.0 + np.abs(g.standard_normal(m))
    B = np.zeros((m, p))
    return Q, R, qd, A, B, l, u


def synth_nodes(first: int, count: int, n: int, m: int, p: int = 8, seed: int = SEED):
    """Stacked node records (Q, R, qd, A, B, l, u) for node ids first .. first+count-1."""
    Q = np.empty((count, n, n)); R = np.empty((count, n, p)); qd = np.empty((count, n))
    A = np.empty((count, m, n)); B = np.zeros((count, m, p))
    l = np.empty((count, m)); u = np.empty((count, m))
    for i in range(count):
        Q[i], R[i], qd[i], A[i], B[i], l[i], u[i] = synth_node(first + i, n, m, p, seed)
    return Q, R, qd, A, B, l, u


def algorithmic_bytes(n: int, m: int) -> int:
    """ALGORITHMIC bytes per node-AVI solve, SURVEY.md section 8(d):
    8 N^2 [M] + 8 N [q] + 16 m [l,u] + 8 N [z0] + 8 N [z] + m [active codes], N = n + m."""
    N = n + m
    return 8 * N * N + 8 * N + 16 * m + 8 * N + 8 * N + m
