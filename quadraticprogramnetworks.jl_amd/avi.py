"""Host-side mirror of src/avi.jl for the hot path: AVI / GAVI carriers, per-node KKT assembly,
pool assembly, GAVI -> AVI conversion, and the solve entry points -- all arithmetic that the
reference hands to PATH (src/avi.jl:64) goes to the HIP engine through the C-ABI.

Dense numpy on the host (the reference is SparseMatrixCSC{Float64,Int32}; sizes here are tiny and
the device kernels are dense).  0-based indices.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum
from typing import Dict, Optional

import numpy as np

from .engine import colmajor, default_engine

INF = np.inf


class StatusCode(IntEnum):          # src/avi.jl:1-6
    SUCCESS = 1
    RAY_TERM = 2
    MAX_ITERS = 3
    FAILURE = 4


@dataclass
class AVI:                           # src/avi.jl:10-16:  (Mz + Nw + o) _|_ l <= z <= u
    M: np.ndarray
    N: np.ndarray
    o: np.ndarray
    l: np.ndarray
    u: np.ndarray


@dataclass
class GAVI:                          # src/avi.jl:18-39
    M: np.ndarray                    # d1 x (d1+d2)
    N: np.ndarray                    # d1 x p
    o: np.ndarray
    l1: np.ndarray
    u1: np.ndarray
    A: np.ndarray                    # d2 x (d1+d2)
    B: np.ndarray                    # d2 x p
    l2: np.ndarray
    u2: np.ndarray


class AVISolveError(RuntimeError):
    """The reference raises error("AVI solve error ...") at src/avi.jl:426."""


def _eng(engine):
    return engine if engine is not None else default_engine(0)


# ---- src/avi.jl:148-156 -------------------------------------------------------------------------
def check_avi_solution(avi: AVI, z, w, tol=1e-6, engine=None):
    q = avi.N @ w + avi.o
    deg, r = _eng(engine).check_avi_batch(colmajor(avi.M), q[None], avi.l[None], avi.u[None],
                                          np.asarray(z, dtype=np.float64)[None], tol=tol)
    return dict(sol_bad=bool(deg[0] > 0), degree=int(deg[0]), r=np.asarray(r[0]))


# ---- src/avi.jl:63-77 ---------------------------------------------------------------------------
def solve_avi(avi: AVI, z0, w, engine=None, opts=None):
    """Drop-in for the PATHSolver.solve_mcp call + post-check.  Returns (z, status, info)."""
    q = avi.N @ w + avi.o
    res = _eng(engine).solve_avi_batch(colmajor(avi.M), q[None], avi.l[None], avi.u[None],
                                       z0=np.asarray(z0, dtype=np.float64)[None], opts=opts)
    return (np.asarray(res["z"][0]), StatusCode(int(res["status"][0])),
            dict(resid=float(res["resid"][0]), pivots=int(res["pivots"][0]), active=np.asarray(res["active"][0])))


# ---- src/avi.jl:113-128 -------------------------------------------------------------------------
def convert(gavi: GAVI) -> AVI:
    d1, d2 = len(gavi.l1), len(gavi.l2)
    p = gavi.N.shape[1]
    M = np.block([[gavi.M, np.zeros((d1, d2))],
                  [gavi.A, -np.eye(d2)],
                  [np.zeros((d2, d1)), np.eye(d2), np.zeros((d2, d2))]])
    N = np.vstack([gavi.N, gavi.B, np.zeros((d2, p))])
    o = np.concatenate([gavi.o, np.zeros(d2), np.zeros(d2)])
    l = np.concatenate([gavi.l1, np.full(d2, -INF), gavi.l2])
    u = np.concatenate([gavi.u1, np.full(d2, INF), gavi.u2])
    return AVI(M, N, o, l, u)


# ---- src/avi.jl:101-111 -------------------------------------------------------------------------
def solve_gavi(gavi: GAVI, z0, w, engine=None, reference_form=False, opts=None):
    """Solve the generalised AVI.  Default: the GAVI's second condition is handed to the kernel
    as GAVI rows (N = d1+d2, no slack block).  reference_form=True goes through convert() and the
    box-AVI of size d1+2*d2 exactly as src/avi.jl:103-110 (used by the parity tests).

    The OSQP pre-projection of src/avi.jl:79-99 only moves the warm start; the pivotal kernel
    starts from its crash basis, so it is not reproduced."""
    d1, d2 = len(gavi.l1), len(gavi.l2)
    z0 = np.asarray(z0, dtype=np.float64)
    if reference_form:
        avi = convert(gavi)
        s = gavi.A @ z0 + gavi.B @ w
        z, status, info = solve_avi(avi, np.concatenate([z0, s]), w, engine=engine, opts=opts)
        return z[:d1 + d2], status, info
    N = d1 + d2
    M = np.vstack([gavi.M, gavi.A])
    q = np.concatenate([gavi.N @ w + gavi.o, gavi.B @ w])
    l = np.concatenate([gavi.l1, gavi.l2])
    u = np.concatenate([gavi.u1, gavi.u2])
    kind = np.concatenate([np.zeros(d1, np.uint8), np.ones(d2, np.uint8)])
    res = _eng(engine).solve_avi_batch(colmajor(M), q[None], l[None], u[None], z0=z0[None],
                                       kind=kind, opts=opts)
    info = dict(resid=float(res["resid"][0]), pivots=int(res["pivots"][0]), active=np.asarray(res["active"][0]))
    return np.asarray(res["z"][0]), StatusCode(int(res["status"][0])), info


# ---- src/avi.jl:205-251 -------------------------------------------------------------------------
def create_labeled_gavi_from_qp(qp_net, pid, solution_graphs: Dict[int, object]):
    """Per-node KKT rows in global-x coordinates.  Z = [x; xi_i; lambda_i; psi_i]:
         M1 = [Q[dvars,:]  0*(-I)  -A_i[:,dvars]'  -A_Si[:,dvars]'],  q1 = q[dvars],  M2 = [A_i; A_Si].
    The string label Dict of :215-242 is pure overhead and is not built."""
    dvars = qp_net.decision_inds(pid)
    n = len(dvars)
    qp = qp_net.qps[pid]
    n_total = qp.f.Q.shape[1]
    rows = [qp_net.constraints[ci].poly.vectorize() for ci in qp.constraint_indices]
    for j in sorted(qp_net.network_edges[pid]):
        rows.append(solution_graphs[j].vectorize())
    if rows:
        A = np.vstack([r[0] for r in rows]); l2 = np.concatenate([r[1] for r in rows]); u2 = np.concatenate([r[2] for r in rows])
    else:
        A = np.zeros((0, n_total)); l2 = np.zeros(0); u2 = np.zeros(0)
    M1 = np.hstack([qp.f.Q[dvars, :], np.zeros((n, n)), -A[:, dvars].T])
    return dict(dvars=dvars, M1=M1, q1=qp.f.q[dvars], M2=A, l2=l2, u2=u2)


# ---- src/avi.jl:305-377 -------------------------------------------------------------------------
def combine_gavis(n, dec_inds, param_inds, labeled_gavis) -> GAVI:
    """Nash pool assembly, reference form: Z = [dvars; xi_i per player; lambda/psi_i per player];
    nd rows "sum_i xi^i_d = 0" on top (:356-367); every z1 bound is +-Inf (:372-373)."""
    nd = len(dec_inds)
    pool = sorted(labeled_gavis.keys())
    xi_dims = {i: labeled_gavis[i]["M1"].shape[0] for i in pool}
    lam_dims = {i: labeled_gavis[i]["M1"].shape[1] - n - xi_dims[i] for i in pool}
    tot_xi = sum(xi_dims.values()); tot_lam = sum(lam_dims.values())
    xi_off, lam_off = {}, {}
    o1, o2 = 0, tot_xi
    for i in pool:
        xi_off[i] = o1; lam_off[i] = o2
        o1 += xi_dims[i]; o2 += lam_dims[i]
    Ms, Ns, qs, As, Bs, ls, us = [], [], [], [], [], [], []
    for i in pool:
        lg = labeled_gavis[i]
        M = lg["M1"]
        Mi = np.zeros((M.shape[0], nd + tot_xi + tot_lam))
        Mi[:, :nd] = M[:, dec_inds]
        Mi[:, nd + xi_off[i]: nd + xi_off[i] + xi_dims[i]] = M[:, n:n + xi_dims[i]]
        Mi[:, nd + lam_off[i]: nd + lam_off[i] + lam_dims[i]] = M[:, n + xi_dims[i]:]
        Ms.append(Mi); Ns.append(M[:, param_inds]); qs.append(lg["q1"])
        As.append(lg["M2"][:, dec_inds]); Bs.append(lg["M2"][:, param_inds]); ls.append(lg["l2"]); us.append(lg["u2"])
    M = np.vstack(Ms); N = np.vstack(Ns); q = np.concatenate(qs)
    A = np.vstack(As); B = np.vstack(Bs); l2 = np.concatenate(ls); u2 = np.concatenate(us)
    top_M = np.zeros((nd, M.shape[1]))
    for i in pool:
        dv = labeled_gavis[i]["dvars"]
        for di, d in enumerate(dec_inds):
            if d in dv:
                top_M[di, nd + xi_off[i] + dv.index(d)] = 1.0
    M = np.vstack([top_M, M]); N = np.vstack([np.zeros((nd, N.shape[1])), N]); o = np.concatenate([np.zeros(nd), q])
    d1 = len(o)
    A = np.hstack([A, np.zeros((A.shape[0], tot_xi + tot_lam))])
    return GAVI(M, N, o, np.full(d1, -INF), np.full(d1, INF), A, B, l2, u2)


def combine_gavis_reduced(n, dec_inds, param_inds, labeled_gavis) -> GAVI:
    """The same pool without the structurally dead xi block (multiplied by 0 at src/avi.jl:244) and
    its nd "sum xi = 0" rows: Z = [dvars; lambda/psi_i per player], d1 = sum n_i.  Requires the
    players' decision sets to be disjoint (sum n_i = nd), which is when the reference form has a
    regular crash basis too."""
    pool = sorted(labeled_gavis.keys())
    nd = len(dec_inds)
    lam_dims = {i: labeled_gavis[i]["M2"].shape[0] for i in pool}
    tot_lam = sum(lam_dims.values())
    if sum(len(labeled_gavis[i]["dvars"]) for i in pool) != nd:
        raise ValueError("reduced pool form needs disjoint decision sets")
    H = np.zeros((nd, nd + tot_lam)); Nn = np.zeros((nd, len(param_inds))); o = np.zeros(nd)
    As, Bs, ls, us = [], [], [], []
    off = 0
    pos = {d: k for k, d in enumerate(dec_inds)}
    for i in pool:
        lg = labeled_gavis[i]
        dv = lg["dvars"]; ni = len(dv)
        rows = [pos[d] for d in dv]
        H[rows, :nd] = lg["M1"][:, dec_inds]
        H[rows, nd + off: nd + off + lam_dims[i]] = lg["M1"][:, n + ni:]
        Nn[rows, :] = lg["M1"][:, param_inds]
        o[rows] = lg["q1"]
        As.append(lg["M2"][:, dec_inds]); Bs.append(lg["M2"][:, param_inds]); ls.append(lg["l2"]); us.append(lg["u2"])
        off += lam_dims[i]
    A = np.hstack([np.vstack(As), np.zeros((tot_lam, tot_lam))]) if tot_lam else np.zeros((0, nd))
    B = np.vstack(Bs) if tot_lam else np.zeros((0, len(param_inds)))
    l2 = np.concatenate(ls) if tot_lam else np.zeros(0); u2 = np.concatenate(us) if tot_lam else np.zeros(0)
    return GAVI(H, Nn, o, np.full(nd, -INF), np.full(nd, INF), A, B, l2, u2)


# ---- the same pool on the device: qpn_assemble_pools ----------------------------------------------
def pool_blocks(n, dec_inds, param_inds, labeled_gavis):
    """The numeric blocks of a pool as qpn_assemble_pools takes them (include/qpn_hip.h): the players' rows stacked in
    pool order (sorted ids, src/avi.jl:319), split into decision / parameter columns, math layout.
    Returns dict(n_i, m_i, dpos, nd, Qd [sn x nd], Qp [sn x p], qd, Ad [sm x nd], Bp [sm x p], l, u)."""
    pool = sorted(labeled_gavis.keys())
    pos = {d: k for k, d in enumerate(dec_inds)}
    n_i = [len(labeled_gavis[i]["dvars"]) for i in pool]
    m_i = [labeled_gavis[i]["M2"].shape[0] for i in pool]
    dpos = [pos[d] for i in pool for d in labeled_gavis[i]["dvars"]]
    Qd = np.vstack([labeled_gavis[i]["M1"][:, dec_inds] for i in pool])
    Qp = np.vstack([labeled_gavis[i]["M1"][:, param_inds] for i in pool])
    qd = np.concatenate([labeled_gavis[i]["q1"] for i in pool])
    Ad = np.vstack([labeled_gavis[i]["M2"][:, dec_inds] for i in pool])
    Bp = np.vstack([labeled_gavis[i]["M2"][:, param_inds] for i in pool])
    l = np.concatenate([labeled_gavis[i]["l2"] for i in pool]); u = np.concatenate([labeled_gavis[i]["u2"] for i in pool])
    return dict(n_i=n_i, m_i=m_i, dpos=dpos, nd=len(dec_inds), Qd=Qd, Qp=Qp, qd=qd, Ad=Ad, Bp=Bp, l=l, u=u)


def assemble_pool_batch(blocks, w, engine=None, form="reduced", qd=None, l=None, u=None):
    """One qpn_assemble_pools call: `blocks` from pool_blocks (shared across the batch); qd / l / u / w may carry a
    leading batch dimension (instances of the pool that differ in their linear terms, bounds or parameters: config 3's
    payoff draws).  Returns (Mc, q, lo, hi, kind) in the ABI layout for solve_avi_batch."""
    b = blocks
    return _eng(engine).assemble_pools(b["n_i"], b["m_i"], b["dpos"], b["nd"], colmajor(b["Qd"]), colmajor(b["Qp"]),
                                       b["qd"] if qd is None else qd, colmajor(b["Ad"]), colmajor(b["Bp"]),
                                       b["l"] if l is None else l, b["u"] if u is None else u, np.asarray(w, dtype=np.float64),
                                       form=form)


# ---- src/avi.jl:382-444 -------------------------------------------------------------------------
def solve_qep(qp_net, player_pool, x, S: Optional[Dict[int, object]] = None, engine=None,
              reference_form=False, settled=None):
    """The AVI step of a level at the current x; returns x_opt.  The reference forms ONE AVI for the whole pool
    (:399-400); that AVI is block diagonal over the connected components of the pool's coupling graph, so the components
    are solved as batches (level_batch.solve_level): single-node components as node records (create_labeled_gavi_from_qp
    fused into the solve kernel), multi-node components through qpn_assemble_pools (combine_gavis on the device; the
    reference form with the xi blocks, the sum-of-xi rows :356-367 and convert :113-128 when asked for or when players
    share decision variables, else the reduced form).  The host mirrors combine_gavis / combine_gavis_reduced above remain
    as the checkers of that kernel (tests/test_gpu_pools.py)."""
    from .level_batch import solve_level
    return solve_level(qp_net, list(player_pool), x, S or {}, engine=engine, reference_form=reference_form, settled=settled)
