"""Host-side mirror of src/qp_processing.jl for the hot path: verify_solution (:57-149),
solve_qp(...; solver=:PATH) (:12-33) and the batch boundary process_qp (:151-241).  The
arithmetic runs on the GPU (qpn_verify_nodes / qpn_solve_avi_batch)."""
from __future__ import annotations

import itertools
from typing import Dict, List

import numpy as np

from .avi import StatusCode, _eng
from .engine import colmajor
from .programs import Poly

INF = np.inf


def node_record(qp, constraints: List[Poly], dec_inds, x):
    """Dense per-node record (Qd, R, qd, Ad, B, l, u, xd, w) of the C-ABI from a QP, its stacked
    constraint rows (src/qp_processing.jl:62-66) and the current point."""
    n_tot = len(x)
    dec = list(dec_inds)
    par = [i for i in range(n_tot) if i not in set(dec)]
    Q = qp.f.Q
    if constraints:
        A = np.vstack([c.vectorize()[0] for c in constraints])
        l = np.concatenate([c.vectorize()[1] for c in constraints])
        u = np.concatenate([c.vectorize()[2] for c in constraints])
    else:
        A = np.zeros((0, n_tot)); l = np.zeros(0); u = np.zeros(0)
    return dict(Qd=Q[np.ix_(dec, dec)], R=Q[np.ix_(dec, par)], qd=qp.f.q[dec], Ad=A[:, dec], B=A[:, par],
                l=l, u=u, xd=np.asarray(x)[dec], w=np.asarray(x)[par], A=A)


def verify_solution(qp, pid, constraints: List[Poly], dec_inds, x, check_convexity=False, tol=1e-4,
                    engine=None):
    """src/qp_processing.jl:57-149 -> dict(solution, lam, e, path)."""
    rec = node_record(qp, constraints, dec_inds, x)
    m = len(rec["l"])
    sol, lam, path = _eng(engine).verify_nodes(colmajor(rec["Qd"])[None], colmajor(rec["R"])[None], rec["qd"][None],
                                               colmajor(rec["Ad"])[None], colmajor(rec["B"])[None],
                                               rec["l"][None], rec["u"][None], rec["xd"][None], rec["w"], tol=tol)
    path = int(path[0])
    msgs = {0: f"Current point is infeasible when using tolerance {tol}.", 1: "Current point is suboptimal",
            4: "Current point is suboptimal (via QP).", 5: "Solving for duals failed."}
    ok = bool(sol[0])
    return dict(solution=ok, lam=(np.asarray(lam[0])[:m].copy() if path in (1, 2, 3, 4) else None),
                e=None if ok else msgs.get(path, ""), path=path)


_VERIFY_MSGS = {0: "Current point is infeasible when using tolerance {tol}.", 1: "Current point is suboptimal",
                4: "Current point is suboptimal (via QP).", 5: "Solving for duals failed."}


def verify_solutions_batched(qp, pid, constraint_lists: List[List[Poly]], dec_inds, x, tol=1e-4, engine=None):
    """verify_solution for MANY constraint stacks of the same node in ONE qpn_verify_nodes call -- the
    sub-piece combinations of src/qp_processing.jl:162-205: the stacks
    share Q, q, x and differ only in the appended child pieces.  Ragged stacks are padded with inert rows
    (0' x in [-inf, inf]: never infeasible, never active, so they enter neither the least-squares system
    of :114-115 nor the fallback of :129-137).  Returns one verify_solution dict per stack, in order."""
    recs = [node_record(qp, cons, dec_inds, x) for cons in constraint_lists]
    if not recs:
        return []
    nb = len(recs)
    n = recs[0]["Qd"].shape[0]
    p = recs[0]["R"].shape[1]
    ms = [len(r["l"]) for r in recs]
    mm = max(max(ms), 1)
    Qc = np.repeat(colmajor(recs[0]["Qd"])[None], nb, axis=0)
    Rc = np.repeat(colmajor(recs[0]["R"])[None], nb, axis=0)
    qd = np.repeat(recs[0]["qd"][None], nb, axis=0)
    xd = np.repeat(recs[0]["xd"][None], nb, axis=0)
    Ad = np.zeros((nb, mm, n)); Bp = np.zeros((nb, mm, p))
    lo = np.full((nb, mm), -INF); hi = np.full((nb, mm), INF)
    for i, r in enumerate(recs):
        Ad[i, :ms[i]] = r["Ad"]; Bp[i, :ms[i]] = r["B"]; lo[i, :ms[i]] = r["l"]; hi[i, :ms[i]] = r["u"]
    sol, lam, path = _eng(engine).verify_nodes(Qc, Rc, qd, colmajor(Ad), colmajor(Bp), lo, hi, xd, recs[0]["w"], tol=tol)
    out = []
    for i in range(nb):
        pth = int(path[i]); ok = bool(sol[i])
        out.append(dict(solution=ok, lam=(np.asarray(lam[i])[:ms[i]].copy() if pth in (1, 2, 3, 4) else None),
                        e=None if ok else _VERIFY_MSGS.get(pth, "").format(tol=tol), path=pth))
    return out


def solve_qp(Q, q, A, l, u, solver="PATH", engine=None):
    """src/qp_processing.jl:12-33 (PATH branch): min 1/2 x'Qx + q'x  s.t. l <= Ax <= u, as the box-MCP
    [Q -A' 0; A 0 -I; 0 I 0] of :16-21 -- sent to the engine in its reduced GAVI-row form."""
    if solver != "PATH":
        raise ValueError("Solver not supported")        # the OSQP branch (:2-11) is out of scope
    Q = np.asarray(Q, dtype=np.float64); A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    n, m = Q.shape[0], A.shape[0]
    M = np.block([[Q, -A.T], [A, np.zeros((m, m))]])
    qq = np.concatenate([q, np.zeros(m)])
    lo = np.concatenate([np.full(n, -INF), l]); hi = np.concatenate([np.full(n, INF), u])
    kind = np.concatenate([np.zeros(n, np.uint8), np.ones(m, np.uint8)])
    res = _eng(engine).solve_avi_batch(colmajor(M), qq[None], lo[None], hi[None], kind=kind)
    if int(res["status"][0]) != StatusCode.SUCCESS:
        raise RuntimeError(f"Solver failure. Status value is {int(res['status'][0])}")     # :30
    return np.asarray(res["z"][0])[:n]


def process_qp(qpn, pid: int, x, S: Dict[int, list], engine=None, exploration_vertices=0):
    """src/qp_processing.jl:151-241 for ONE node.  S maps child id -> list of Poly pieces.  For every combination of the
    children's pieces (:162-169) the node is verified (:187, one batched call); if it is optimal under all of them and a
    solution graph is wanted (:158), each combination's pieces come from the device kernels (masks -> recipes -> local pieces
    with the multipliers eliminated) and are put together by `combine_at` (:219, :260-291).  The outer loop does not call this
    per node: level_batch.process_level serves all nodes of a level with the same calls (src/algorithm.jl:44-52)."""
    from .level_batch import process_level
    return process_level(qpn, [pid], x, S, engine=engine, exploration_vertices=exploration_vertices)[0]


def _halfspace_complements(P: Poly, cols, x, tol=1e-6):
    """complement(p::Poly) (src/sets.jl:918-930) over the compressed columns `cols`: one open half-space per finite bound of
    every row.  Returns (all of them, the ones whose closure contains x)."""
    A = P.block(cols)
    xs = np.asarray(x, dtype=np.float64)[cols]
    every, near = [], []
    for i in range(A.shape[0]):
        ax = float(A[i] @ xs)
        if np.isfinite(P.l[i]):                             # a'x < l (the relation complemented: closed -> open)
            H = Poly(A[i:i + 1], [-INF], [P.l[i]], normalise=False, open_hi=[not P.open_lo[i]])
            every.append(H)
            if ax <= P.l[i] + tol:
                near.append(H)
        if np.isfinite(P.u[i]):                             # a'x > u
            H = Poly(A[i:i + 1], [P.u[i]], [INF], normalise=False, open_lo=[not P.open_hi[i]])
            every.append(H)
            if ax >= P.u[i] - tol:
                near.append(H)
    return every, near


def _combine_products(regions: List[List[Poly]], solutions: List[List[Poly]], x):
    """The candidate products of combine(...) for one node: (cols, ncols, products), every product a Poly over the compressed
    columns `cols` whose closure contains x; raises RuntimeError on the reference's size guard (src/qp_processing.jl:281-285)."""
    x = np.asarray(x, dtype=np.float64)
    polys = [P for R in regions for P in R] + [P for Sg in solutions for P in Sg]
    cols = np.unique(np.concatenate([P.support() for P in polys])) if polys else np.zeros(0, np.int64)
    ncols = polys[0].ncols if polys else len(x)
    widths, cands = [], []
    for R, Sg in zip(regions, solutions):
        every, near = [], []
        for P in R:
            e, nr = _halfspace_complements(P, cols, x)
            every += e; near += nr
        widths.append(len(Sg) + len(every))
        mine = [(False, Poly(P.block(cols), P.l, P.u, normalise=False, open_lo=P.open_lo, open_hi=P.open_hi)) for P in Sg]
        cands.append(mine + [(True, H) for H in near])
    if len(widths) > 3 and sum(widths) > 20:                # :281-285
        raise RuntimeError("Too many solutions to combine.")
    xs = x[cols]
    prods = []
    for choice in itertools.product(*cands):
        if all(c[0] for c in choice):                       # complement pieces only: the redzone (src/intersection.jl:124)
            continue
        A = np.vstack([c[1].A for c in choice]); l = np.concatenate([c[1].l for c in choice]); u = np.concatenate([c[1].u for c in choice])
        olo = np.concatenate([c[1].open_lo for c in choice]); ohi = np.concatenate([c[1].open_hi for c in choice])
        if not Poly(A, l, u, normalise=False).contains(xs):                         # central_point in closure(...), src/intersection.jl:74
            continue
        prods.append(Poly(A, l, u, normalise=False, open_lo=olo, open_hi=ohi))
    return cols, ncols, prods


def combine_many(jobs, x, engine, tol=1e-4):
    """combine(regions, solutions, x) (src/qp_processing.jl:260-291) for ALL nodes of a level that are optimal under SEVERAL
    combinations of their children's pieces (x sits on a kink of a child's solution graph).  Per node

        S = intersection over the combinations i of ( S_i  union  complement(R_i) ),

    R_i = the intersection of combination i's child pieces, S_i = the node's solution pieces under it.  The reference
    enumerates the products lazily (IntersectionRoot, src/intersection.jl:55-138) and keeps a product when the current point
    lies in its closure and it is not empty (:74, :83), skipping products made of complement pieces only (the "redzone",
    :124).  Only candidates whose closure contains x can survive, so they alone are expanded; the emptiness questions of all
    nodes go to the engine as ONE batch of LPs (polyhedra.isempty_slack_batch: `isempty`, src/sets.jl:647-655 -> `exemplar`,
    :591-642, open bounds included).  jobs: list of (regions, solutions); returns per job the list of pieces (global
    coordinates) or the RuntimeError of the reference's size guard (:281-285) for the caller to turn into failed = true."""
    from .polyhedra import isempty_slack_batch
    prep, flat, owner = [], [], []
    for k, (regions, solutions) in enumerate(jobs):
        try:
            cols, ncols, prods = _combine_products(regions, solutions, x)
        except RuntimeError as err:
            prep.append(err)
            continue
        prep.append((cols, ncols, prods))
        flat += prods; owner += [k] * len(prods)
    empty = isempty_slack_batch(flat, engine, tol=tol) if flat else []
    out = []
    for k, pr in enumerate(prep):
        if isinstance(pr, RuntimeError):
            out.append(pr)
            continue
        cols, ncols, _ = pr
        out.append([Poly.from_local(ncols, cols, P.A, P.l, P.u, normalise=False, open_lo=P.open_lo, open_hi=P.open_hi)
                    for P, e, o in zip(flat, empty, owner) if o == k and not e])
    return out


def combine_at(regions: List[List[Poly]], solutions: List[List[Poly]], x, engine, tol=1e-4):
    """combine_many for one node; raises the size guard's RuntimeError."""
    got = combine_many([(regions, solutions)], x, engine, tol=tol)[0]
    if isinstance(got, RuntimeError):
        raise got
    return got


def local_recipe_count(qpn, pid: int, x, S: Dict[int, list], engine=None):
    """How many local pieces the node's solution graph has AT x: over every combination of the children's pieces for which
    the node is optimal (src/qp_processing.jl:162-205), the number of recipes compatible with the active-set masks of the
    node's own GAVI (process_solution_graph, src/avi.jl:447-477 -> comp_indices -> all_Ks, src/avi_solutions.jl:200-215):
    each of them is a non-empty piece containing x (local_piece, :400-496).  A lower bound of what the reference's graph
    enumeration collects (its pieces start from these and grow by exploration), computable on the hot path alone."""
    from .avi import GAVI
    from .avi_solutions import comp_indices
    qp = qpn.qps[pid]
    base = [qpn.constraints[c].poly for c in qp.constraint_indices]
    dec = qpn.decision_inds(pid)
    children = sorted(qpn.network_edges[pid])
    combos = list(itertools.product(*[range(len(S[j])) for j in children])) if children else [()]
    stacks = [base + [S[j][ji] for j, ji in zip(children, combo)] for combo in combos]
    rets = verify_solutions_batched(qp, pid, stacks, dec, x, engine=engine)
    total = 0
    for cons, r in zip(stacks, rets):
        if not r["solution"]:
            continue
        rec = node_record(qp, cons, dec, x)
        n, m = len(dec), len(rec["l"])
        g = GAVI(np.hstack([rec["Qd"], -rec["Ad"].T]), rec["R"], rec["qd"], np.full(n, -INF), np.full(n, INF),
                 np.hstack([rec["Ad"], np.zeros((m, m))]), rec["B"], rec["l"], rec["u"])
        mask = comp_indices(g, np.concatenate([rec["xd"], r["lam"]]), rec["w"], engine=engine)
        total += int(np.prod([bin(int(v)).count("1") for v in mask]))
    return total
