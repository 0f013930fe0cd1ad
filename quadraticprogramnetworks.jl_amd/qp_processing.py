"""Host-side mirror of src/qp_processing.jl for the hot path: verify_solution (:57-149),
solve_qp(...; solver=:PATH) (:12-33) and the batch boundary process_qp (:151-241).  The
arithmetic runs on the GPU (qpn_verify_nodes / qpn_solve_avi_batch)."""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional

import numpy as np

from .avi import StatusCode, _eng
from .avi_solutions import solution_graph_pieces
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
    """src/qp_processing.jl:151-241.  S maps child id -> list of Poly pieces.  For every combination of the children's
    pieces (:162-169) the node is verified (:187, one batched call).  Solution-graph generation (:193-198, :231): the pieces
    come from the device kernels (avi_solutions.solution_graph_pieces: masks -> qpn_recipes_from_masks -> qpn_local_pieces)
    for leaves and for every sub-piece combination of an inner node; an inner node returns each combination's pieces
    intersected with the child pieces they were derived under -- the first generation of what combine(...) (:219,
    IntersectionRoot, polyhedral: out of scope) would enumerate, without its exploration."""
    qp = qpn.qps[pid]
    base = [qpn.constraints[c].poly for c in qp.constraint_indices]
    dec_inds = qpn.decision_inds(pid)
    gen = (pid not in qpn.network_depth_map[1]) or qpn.options.gen_solution_map
    children = sorted(qpn.network_edges[pid])
    if children:
        cards = [range(len(S[j])) for j in children]
        if any(len(c) < 1 for c in cards):
            raise RuntimeError("Solution graphs were not properly populated.")
        # every combination is verified (the reference maps over all of them, :171-205, and then reports the
        # first failure in product order, :206-216): ONE batched call instead of one call per combination
        combos = list(itertools.product(*cards))
        rets = verify_solutions_batched(qp, pid, [base + [S[j][ji] for j, ji in zip(children, combo)] for combo in combos],
                                        dec_inds, x, engine=engine)
        for combo, ret in zip(combos, rets):
            if not ret["solution"]:
                return dict(solution=False, e=ret["e"], failed=False,
                            subpiece_assignments={j: ji for j, ji in zip(children, combo)})
        S_out = None
        if gen:
            # :193-198 per combination: the node's GAVI under this combination's child pieces -> device pieces
            S_out = []
            for combo, ret in zip(combos, rets):
                cons = base + [S[j][ji] for j, ji in zip(children, combo)]
                rec = node_record(qp, cons, dec_inds, x)
                pieces = solution_graph_pieces(qp.f.Q, qp.f.q, rec["A"], rec["l"], rec["u"], dec_inds, np.asarray(x), ret["lam"],
                                               engine=engine)
                child = [S[j][ji] for j, ji in zip(children, combo)]
                for P in pieces:
                    Ap, lp, up = P.vectorize()
                    rows = [c.vectorize() for c in child]
                    S_out.append(Poly(np.vstack([Ap] + [r[0] for r in rows]), np.concatenate([lp] + [r[1] for r in rows]),
                                      np.concatenate([up] + [r[2] for r in rows])))
            if len(S_out) == 0:
                raise RuntimeError("This shouldn't happen. Solution graph is empty.")
        return dict(solution=True, S=S_out, failed=False)
    ret = verify_solution(qp, pid, base, dec_inds, x, engine=engine)
    if not ret["solution"]:
        return dict(solution=False, e=ret["e"], failed=False, subpiece_assignments={})
    S_out = None
    if gen:
        rec = node_record(qp, base, dec_inds, x)
        S_out = solution_graph_pieces(qp.f.Q, qp.f.q, rec["A"], rec["l"], rec["u"], dec_inds, np.asarray(x), ret["lam"],
                                      engine=engine)
        if len(S_out) == 0:
            raise RuntimeError("This shouldn't happen. Solution graph is empty.")
    return dict(solution=True, S=S_out, failed=False)


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
