"""Independent numpy solvers used only to cross-check the oracle (different algorithms:
semismooth Newton on the natural map, and brute-force active-set enumeration)."""
from __future__ import annotations

import itertools

import numpy as np


def natural_map(M, q, l, u, kind, z):
    F = M @ z + q
    p = np.where(kind == 1, F, z)
    d = np.where(kind == 1, z, F)
    return p - np.clip(p - d, l, u), p, d


def solve_newton(M, q, l, u, kind=None, z0=None, iters=200, tol=1e-11):
    """Semismooth Newton (primal-dual active set) on the natural map; converges for the
    strongly monotone instances the tests generate.  Returns (z, resid)."""
    N = len(q)
    kind = np.zeros(N, dtype=np.uint8) if kind is None else np.asarray(kind)
    z = np.zeros(N) if z0 is None else np.array(z0, dtype=float)
    z = np.where((kind == 0) & np.isfinite(l) & (z < l), l, z)
    z = np.where((kind == 0) & np.isfinite(u) & (z > u), u, z)
    I = np.eye(N)
    for _ in range(iters):
        phi, p, d = natural_map(M, q, l, u, kind, z)
        if np.max(np.abs(phi)) < tol:
            break
        t = p - d
        inside = (t > l) & (t < u)
        J = np.empty((N, N))
        rhs = np.empty(N)
        for i in range(N):
            dp = M[i] if kind[i] == 1 else I[i]
            dd = I[i] if kind[i] == 1 else M[i]
            if inside[i]:
                J[i] = dd          # phi_i = d_i
                rhs[i] = -d[i]
            else:
                J[i] = dp          # phi_i = p_i - bound
                rhs[i] = -(p[i] - (l[i] if t[i] <= l[i] else u[i]))
        try:
            dz = np.linalg.solve(J, rhs)
        except np.linalg.LinAlgError:
            dz = np.linalg.lstsq(J, rhs, rcond=None)[0]
        # backtracking on ||phi||_inf (plain primal-dual active-set steps can cycle)
        t = 1.0
        f0 = np.max(np.abs(phi))
        while t > 1e-4:
            f1 = np.max(np.abs(natural_map(M, q, l, u, kind, z + t * dz)[0]))
            if f1 < (1 - 1e-4 * t) * f0:
                break
            t *= 0.5
        z = z + t * dz
    phi, _, _ = natural_map(M, q, l, u, kind, z)
    return z, float(np.max(np.abs(phi)))


def solve_enumerate(M, q, l, u, kind=None, tol=1e-9):
    """Brute force over active-set patterns (N <= ~10): every row is at-lower / inside / at-upper."""
    N = len(q)
    kind = np.zeros(N, dtype=np.uint8) if kind is None else np.asarray(kind)
    I = np.eye(N)
    sols = []
    opts = []
    for i in range(N):
        o = [1]
        if np.isfinite(l[i]):
            o.append(0)
        if np.isfinite(u[i]) and u[i] != l[i]:
            o.append(2)
        opts.append(o)
    for pat in itertools.product(*opts):
        J = np.empty((N, N))
        rhs = np.empty(N)
        for i, s in enumerate(pat):
            dp = M[i] if kind[i] == 1 else I[i]
            dd = I[i] if kind[i] == 1 else M[i]
            cp = q[i] if kind[i] == 1 else 0.0
            cd = 0.0 if kind[i] == 1 else q[i]
            if s == 1:
                J[i] = dd; rhs[i] = -cd
            elif s == 0:
                J[i] = dp; rhs[i] = l[i] - cp
            else:
                J[i] = dp; rhs[i] = u[i] - cp
        try:
            z = np.linalg.solve(J, rhs)
        except np.linalg.LinAlgError:
            continue
        phi, p, d = natural_map(M, q, l, u, kind, z)
        if np.max(np.abs(phi)) < tol:
            if not any(np.allclose(z, s2, atol=1e-7) for s2 in sols):
                sols.append(z)
    return sols
