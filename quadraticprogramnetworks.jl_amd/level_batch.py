"""A whole LEVEL of a QPNet as batches: the per-node map of process_qp (src/algorithm.jl:44-52), the sub-piece combinations
of src/qp_processing.jl:162-205 and the level's AVI step (solve_qep, src/avi.jl:382-444) served by O(1) device calls per
level and outer iteration, whatever the number of nodes.

The reference maps process_qp over the nodes of a level one by one and then forms ONE AVI for all of them
(src/avi.jl:399-400).  That AVI is block diagonal whenever the level's coupling graph is disconnected -- 5 000 followers
that read only their own leader are 5 000 independent node-AVIs, not one AVI of 320 000 unknowns -- so here

  * `components` splits the level's pool by the coupling graph (shared decision variables, or a player's KKT rows reading
    another player's decision variables: src/avi.jl:335-340 puts exactly those columns into M instead of N);
  * single-node components of equal shape are ONE batch of node records (qpn_solve_nodes / a resident qpn_nodes handle for
    nodes without children, whose records never change between outer iterations);
  * multi-node components of equal pool shape are ONE batch of pool AVIs (qpn_assemble_pools + qpn_solve_avi_batch);
  * `process_level` verifies every node of the level under every combination of its children's pieces in one
    qpn_verify_nodes call per record shape, and makes the solution-graph pieces of all optimal nodes with one
    comp_indices / recipes / pieces call each (src/avi.jl:447-477, src/avi_solutions.jl:200-215, :400-496).

A node's parameters are LOCAL here: only the variables its rows actually read (R, B have one column per such variable),
gathered from the iterate per sweep -- the reference's N = M[:, param_inds] over all n_total - n other variables
(src/avi.jl:340) is structurally zero outside them.

Everything numeric goes through the engine (the HIP library; the tests' CPU twin serves the same interface).
"""
from __future__ import annotations

import itertools
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

from .programs import Poly, _positions

INF = np.inf
# A local piece enters a node's solution graph only when it contains the current point -- within MEMBER_TOL, tighter than
# verify_solution's accept band (1e-4, src/qp_processing.jl:57, :119) and far tighter than its feasibility tolerance (1e-3, :86).  comp_indices classifies with 1e-2 (src/avi_solutions.jl:511), so a row up to 1e-2 away
# from its bound still spawns the recipe "at the bound"; the reference keeps every such piece that is non-empty
# (src/avi_solutions.jl:247-249).  A piece the point misses by more than 1e-3 fails the PARENT's verify by construction (:86-89:
# infeasible), the parent's solve_qep then moves onto it, the next sweep finds the mirror image, and the loop ends in the
# reference's own "Cycling detected" (observed: one pair in 60 at n = m = 16).  Such pieces say nothing about optimality AT the
# point, so they are left out.  The tolerance has to sit BELOW the accept band: two neighbouring pieces' exact minimisers can lie
# 1e-4 apart (the parent optimal under B a hair inside B, and within 1e-3 of A's boundary): with a membership test as loose as
# verify's feasibility test each of the two points counts as a member of the other piece without being optimal there, and the
# parent is sent back and forth by 1e-4 for ever (observed: one pair in ~1 000 at n = m = 16).
MEMBER_TOL = 1e-5
CODE_TOL = 1e-4          # _refine_row_codes: a row keeps a code whose own condition the point meets within verify's tolerance (:57)
ROW_PAD = 16            # constraint rows of a record batch are padded with inert rows to a multiple of this (one MFMA tile)


# ---------------------------------------------------------------------------------------------------------------------
# node records
# ---------------------------------------------------------------------------------------------------------------------
def _static(qpn, pid: int) -> dict:
    """What a node's records owe to the net alone: decision indices, Q blocks, the base constraint rows."""
    cache = qpn.__dict__.setdefault("_node_static", {})
    st = cache.get(pid)
    if st is not None:
        return st
    qp = qpn.qps[pid]
    dec = np.asarray(qpn.decision_inds(pid), dtype=np.int64)
    base = [qpn.constraints[c].poly for c in qp.constraint_indices]
    f = qp.f
    supp = [f.row_support(dec)] + [P.support() for P in base]
    supp = np.unique(np.concatenate(supp)) if supp else np.zeros(0, np.int64)
    par = np.setdiff1d(supp, dec, assume_unique=True)
    st = dict(dec=dec, base=base, par=par, Qd=f.block(dec, dec), qd=f.q_at(dec),
              Ad=(np.vstack([P.block(dec) for P in base]) if base else np.zeros((0, dec.size))),
              l=(np.concatenate([P.l for P in base]) if base else np.zeros(0)),
              u=(np.concatenate([P.u for P in base]) if base else np.zeros(0)),
              # the blocks on the static parameters: what a record under child pieces that bring no new column keeps as it is
              R=f.block(dec, par), B=(np.vstack([P.block(par) for P in base]) if base else np.zeros((0, par.size))),
              reads=np.union1d(dec, par))
    cache[pid] = st
    return st


def _subtree_cols(qpn, pid: int) -> np.ndarray:
    """Every variable process_qp(pid) can read: the decision variables and static parameters of the node and of all nodes
    below it (the children's pieces are functions of exactly those)."""
    cache = qpn.__dict__.setdefault("_subtree_cols", {})
    got = cache.get(pid)
    if got is None:
        st = _static(qpn, pid)
        parts = [st["dec"], st["par"]] + [_subtree_cols(qpn, j) for j in sorted(qpn.network_edges[pid])]
        got = cache[pid] = np.unique(np.concatenate(parts))
    return got


def node_record(qpn, pid: int, child_polys: Sequence[Poly] = ()) -> dict:
    """Dense record of one node under one choice of its children's pieces (src/qp_processing.jl:62-66: the constraint
    stack is [base; child pieces]) with LOCAL parameters: dict(pid, dec, par, Qd, R, qd, Ad, B, l, u) in math layout."""
    st = _static(qpn, pid)
    dec = st["dec"]
    f = qpn.qps[pid].f
    if not child_polys:
        return dict(pid=pid, dec=dec, par=st["par"], Qd=st["Qd"], R=st["R"], qd=st["qd"], Ad=st["Ad"], B=st["B"], l=st["l"], u=st["u"])
    supp = [P.support() for P in child_polys]
    if all(_positions(st["reads"], c)[1].all() for c in supp):
        # the usual case: the children's pieces live on variables the node reads anyway (its own and its static parameters)
        par = st["par"]
        return dict(pid=pid, dec=dec, par=par, Qd=st["Qd"], R=st["R"], qd=st["qd"],
                    Ad=np.vstack([st["Ad"]] + [P.block(dec) for P in child_polys]),
                    B=np.vstack([st["B"]] + [P.block(par) for P in child_polys]),
                    l=np.concatenate([st["l"]] + [P.l for P in child_polys]),
                    u=np.concatenate([st["u"]] + [P.u for P in child_polys]))
    if child_polys:
        extra = np.unique(np.concatenate(supp))
        par = np.union1d(st["par"], np.setdiff1d(extra, dec, assume_unique=True))
        Ad = np.vstack([st["Ad"]] + [P.block(dec) for P in child_polys])
        l = np.concatenate([st["l"]] + [P.l for P in child_polys])
        u = np.concatenate([st["u"]] + [P.u for P in child_polys])
        cons = list(st["base"]) + list(child_polys)
    else:
        par, Ad, l, u, cons = st["par"], st["Ad"], st["l"], st["u"], st["base"]
    R = f.block(dec, par)
    B = np.vstack([P.block(par) for P in cons]) if cons else np.zeros((0, par.size))
    return dict(pid=pid, dec=dec, par=par, Qd=st["Qd"], R=R, qd=st["qd"], Ad=Ad, B=B, l=l, u=u)


def free_equalities(rec: dict) -> dict:
    """The same node with its equality rows (l = u) moved into the free block: an equality row's multiplier mu is a FREE variable
    of the node's AVI (src/avi.jl:113-128: the row a'x + b'w = e pairs with an unbounded multiplier), so z = [x; mu_E; lambda_I]
    with the free block [[Qd, -A_E'], [A_E, 0]] (right-hand side [qd; -e], parameter rows [R; B_E]) and the remaining rows
    [A_I 0] is the identical complementarity system.  The fused node kernels eliminate the free block without pivoting -- in this
    order: x first (Qd, definite), then mu (its block has become A_E Qd^-1 A_E', definite for independent rows) -- and run the
    complementary pivoting on the inequality rows alone; a record that keeps an equality row is declined by every one of them
    (the multiplier would have to be crashed in) and falls to the general kernel.  Every parent of a net has such rows: the
    stationarity rows of its children's pieces.  `nx` = the number of leading free variables that are decision variables."""
    l, u = rec["l"], rec["u"]
    eq = np.isfinite(l) & (l == u)
    if not eq.any():
        return rec
    E = np.nonzero(eq)[0]; I = np.nonzero(~eq)[0]
    n, ne = rec["Qd"].shape[0], E.size
    AE, BE = rec["Ad"][E], rec["B"][E]
    out = dict(rec)
    out["Qd"] = np.block([[rec["Qd"], -AE.T], [AE, np.zeros((ne, ne))]])
    out["R"] = np.vstack([rec["R"], BE])
    out["qd"] = np.concatenate([rec["qd"], -l[E]])
    out["Ad"] = np.hstack([rec["Ad"][I], np.zeros((I.size, ne))])
    out["B"] = rec["B"][I]
    out["l"], out["u"] = l[I], u[I]
    out["nx"] = n
    return out


class RecordBatch:
    """Records of equal shape (n, padded m, padded p) stacked in the ABI layout (column-major per item).  n counts the free
    block (Qd is n x n); its first nx entries are the node's decision variables (nx < n after free_equalities)."""

    def __init__(self, recs: List[dict], where: List[int], n: int, m: int, p: int):
        nb = len(recs)
        self.where = list(where)                     # positions of these records in the caller's list
        self.n, self.m, self.p = n, m, p
        self.m_true = np.array([len(r["l"]) for r in recs], dtype=np.int64)
        self.Qc = np.zeros((nb, n, n)); self.Rc = np.zeros((nb, p, n)); self.qd = np.zeros((nb, n))
        self.Ac = np.zeros((nb, n, m)); self.Bc = np.zeros((nb, p, m))
        self.l = np.full((nb, m), -INF); self.u = np.full((nb, m), INF)      # missing rows: 0'x in (-inf, inf) -- inert
        self.nx = recs[0]["dec"].size
        self.dec = np.zeros((nb, self.nx), dtype=np.int64)
        self.par = np.full((nb, p), -1, dtype=np.int64)                      # missing parameters: a zero column, w = 0
        mis = {len(r["l"]) for r in recs}; pis = {r["par"].size for r in recs}
        if len(mis) == 1 and len(pis) == 1:
            # the usual batch -- one row count, one parameter count: stacked and transposed in one go each
            mi, pi = mis.pop(), pis.pop()
            self.Qc[:] = np.stack([r["Qd"] for r in recs]).transpose(0, 2, 1)
            self.qd[:] = np.stack([r["qd"] for r in recs])
            self.dec[:] = np.stack([r["dec"] for r in recs])
            if pi:
                self.Rc[:, :pi] = np.stack([r["R"] for r in recs]).transpose(0, 2, 1)
                self.par[:, :pi] = np.stack([r["par"] for r in recs])
            if mi:
                self.Ac[:, :, :mi] = np.stack([r["Ad"] for r in recs]).transpose(0, 2, 1)
                self.l[:, :mi] = np.stack([r["l"] for r in recs]); self.u[:, :mi] = np.stack([r["u"] for r in recs])
                if pi:
                    self.Bc[:, :pi, :mi] = np.stack([r["B"] for r in recs]).transpose(0, 2, 1)
            recs = []
        for b, r in enumerate(recs):
            mi, pi = len(r["l"]), r["par"].size
            self.Qc[b] = r["Qd"].T
            self.Rc[b, :pi] = r["R"].T
            self.qd[b] = r["qd"]
            self.Ac[b, :, :mi] = r["Ad"].T
            self.Bc[b, :pi, :mi] = r["B"].T
            self.l[b, :mi] = r["l"]; self.u[b, :mi] = r["u"]
            self.dec[b] = r["dec"]; self.par[b, :pi] = r["par"]
        self.handle = None                           # resident copy (engine.upload_nodes), when the records are static
        self.handle_engine = None

    def __len__(self):
        return len(self.where)

    def gather(self, x):
        x = np.asarray(x, dtype=np.float64)
        xd = x[self.dec]
        w = np.where(self.par >= 0, x[np.maximum(self.par, 0)], 0.0)
        return xd, w

    def resident(self, engine):
        """The records as a resident handle of `engine` (made once), or None when the engine has no such thing."""
        if not hasattr(engine, "upload_nodes"):
            return None
        if self.handle is None or self.handle_engine is not engine:
            self.handle = engine.upload_nodes(self.Qc, self.Rc, self.qd, self.Ac, self.Bc, self.l, self.u)
            self.handle_engine = engine
        return self.handle


def batch_records(recs: List[dict], row_pad: int = ROW_PAD) -> List[RecordBatch]:
    """Group records by (n, m rounded up to a multiple of row_pad); p is padded to the group's widest record (>= 1)."""
    groups: Dict[tuple, List[int]] = {}
    for i, r in enumerate(recs):
        m = len(r["l"])
        mp = max(row_pad, -(-m // row_pad) * row_pad) if m else 0
        groups.setdefault((r["Qd"].shape[0], mp, r["dec"].size), []).append(i)
    out = []
    for (n, mp, _nx), idx in sorted(groups.items()):
        p = max(1, max(recs[i]["par"].size for i in idx))
        out.append(RecordBatch([recs[i] for i in idx], idx, n, mp, p))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the level's pool split into connected components
# ---------------------------------------------------------------------------------------------------------------------
def components(qpn, players: Sequence[int], assign: Optional[Dict[int, Poly]] = None) -> List[List[int]]:
    """Connected components of the level's coupling graph.  Players i and j are coupled when they share a decision
    variable or when the KKT rows of one (Q_i[dvars_i, :], its base rows, its children's chosen pieces) read a decision
    variable of the other: those are the columns src/avi.jl:335-340 keeps in M.  Components come back as sorted id lists,
    ordered by their smallest id."""
    assign = assign or {}
    players = sorted(players)
    owner: Dict[int, int] = {}
    parent = {i: i for i in players}

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    decs = {}
    for i in players:
        st = _static(qpn, i)
        decs[i] = st["dec"]
        for v in st["dec"].tolist():
            o = owner.get(v)
            if o is None:
                owner[v] = i
            else:
                ra, rb = find(o), find(i)
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
    for i in players:
        reads = [_static(qpn, i)["par"]]
        for j in sorted(qpn.network_edges[i]):
            if j in assign:
                reads.append(assign[j].support())
        for v in np.unique(np.concatenate(reads)).tolist():
            o = owner.get(v)
            if o is not None:
                ra, rb = find(o), find(i)
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
    comps: Dict[int, List[int]] = {}
    for i in players:
        comps.setdefault(find(i), []).append(i)
    return [sorted(c) for _, c in sorted(comps.items())]


def _pool_blocks_local(qpn, pool: List[int], assign: Dict[int, Poly]):
    """The numeric blocks of ONE multi-node component as qpn_assemble_pools takes them (avi.pool_blocks), with the
    component's own decision set and local parameters."""
    stat = {i: _static(qpn, i) for i in pool}
    dec = np.unique(np.concatenate([stat[i]["dec"] for i in pool]))
    cons = {i: list(stat[i]["base"]) + [assign[j] for j in sorted(qpn.network_edges[i]) if j in assign] for i in pool}
    reads = [stat[i]["par"] for i in pool] + [P.support() for i in pool for P in cons[i][len(stat[i]["base"]):]]
    par = np.setdiff1d(np.unique(np.concatenate(reads)), dec, assume_unique=True)
    pos = {int(d): k for k, d in enumerate(dec.tolist())}
    n_i = [stat[i]["dec"].size for i in pool]
    m_i = [sum(len(P) for P in cons[i]) for i in pool]
    dpos = [pos[int(d)] for i in pool for d in stat[i]["dec"].tolist()]
    f = {i: qpn.qps[i].f for i in pool}
    zero_d = np.zeros((0, dec.size)); zero_p = np.zeros((0, par.size))
    Qd = np.vstack([f[i].block(stat[i]["dec"], dec) for i in pool])
    Qp = np.vstack([f[i].block(stat[i]["dec"], par) for i in pool])
    qd = np.concatenate([stat[i]["qd"] for i in pool])
    Ad = np.vstack([zero_d] + [P.block(dec) for i in pool for P in cons[i]])
    Bp = np.vstack([zero_p] + [P.block(par) for i in pool for P in cons[i]])
    l = np.concatenate([np.zeros(0)] + [P.l for i in pool for P in cons[i]])
    u = np.concatenate([np.zeros(0)] + [P.u for i in pool for P in cons[i]])
    return dict(n_i=n_i, m_i=m_i, dpos=dpos, nd=int(dec.size), Qd=Qd, Qp=Qp, qd=qd, Ad=Ad, Bp=Bp, l=l, u=u, dec=dec, par=par)


# ---------------------------------------------------------------------------------------------------------------------
# solve_qep for a level: src/avi.jl:382-444 over the components
# ---------------------------------------------------------------------------------------------------------------------
def _leaf_batches(qpn, players, engine, free_eq=False):
    """Record batches of the childless nodes among `players`: they depend on the net alone, so they are built once per
    (net, player set) and keep a resident handle.  free_eq: the solve's form of the records (free_equalities)."""
    cache = qpn.__dict__.setdefault("_leaf_batches", {})
    key = (tuple(players), bool(free_eq))
    got = cache.get(key)
    if got is None:
        recs = [node_record(qpn, i) for i in players]
        if free_eq:
            recs = [free_equalities(r) for r in recs]
        got = cache[key] = (recs, batch_records(recs))
    return got


def solve_level(qpn, players: Sequence[int], x, assign: Optional[Dict[int, Poly]] = None, engine=None, reference_form=False,
                settled: Optional[set] = None):
    """solve_qep (src/avi.jl:382-444) for a whole level: x_opt with every component's decision block replaced by the
    solution of its AVI.  Raises avi.AVISolveError when any component's solve does not end in SUCCESS (:426).

    `settled` (ids of players verify_solution has just accepted): a component ALL of whose players are settled is left where it
    is.  The reference re-solves it with the rest of the level, under its children's FIRST pieces (src/algorithm.jl:54, :95); the
    point passes verify_solution under every piece combination, so that solve returns it again -- up to the 1e-4 band of the
    accept tests, inside which two pieces' exact minimisers can differ: re-solved sweep after sweep while OTHER components are
    still moving, such a component jumps between them and the level ends in "Cycling detected", although the component on its
    own (or the net sharded by cluster, sharding.solve_sharded) is done.  With the split into components the engine knows
    which blocks of the level's AVI are already at rest."""
    from .avi import AVISolveError, StatusCode, _eng
    from .engine import colmajor
    eng = _eng(engine)
    assign = assign or {}
    x = np.asarray(x, dtype=np.float64)
    x_opt = x.copy()
    comps = components(qpn, players, assign)
    if settled:
        comps = [c for c in comps if not all(i in settled for i in c)]
    singles = [c[0] for c in comps if len(c) == 1]
    multis = [c for c in comps if len(c) > 1]
    bad = []
    # ---- single-node components: node records, one call per record shape
    leaf = [i for i in singles if not qpn.network_edges[i]]
    inner = [i for i in singles if qpn.network_edges[i]]
    work = []
    take = None
    if leaf and not reference_form:
        # ONE resident batch per level for all its childless players (built once per net): every record is solved -- one launch
        # whatever the subset -- and only the players asked for are written back (a batch per subset would upload a new copy of
        # the records whenever the set of unsettled players changes)
        universe = [i for i in sorted(players) if not qpn.network_edges[i]]
        recs_u, batches_u = _leaf_batches(qpn, universe, eng, free_eq=True)
        wanted = set(leaf)
        take = {id(b): np.array([recs_u[i]["pid"] in wanted for i in b.where]) for b in batches_u}
        work += [(b, True) for b in batches_u if take[id(b)].any()]
    elif leaf:
        inner = sorted(inner + leaf)
    if inner:
        recs = [node_record(qpn, i, [assign[j] for j in sorted(qpn.network_edges[i]) if j in assign]) for i in inner]
        if reference_form:
            work += [(r, None) for r in recs]
        else:
            work += [(b, False) for b in batch_records([free_equalities(r) for r in recs])]
    for item, static in work:
        if static is None:
            _solve_single_reference_form(item, x, x_opt, eng, bad)
            continue
        b = item
        xd, w = b.gather(x)
        z0 = np.zeros((len(b), b.n + b.m)); z0[:, :b.nx] = xd                       # duals cold, :404
        h = b.resident(eng) if static else None
        if h is not None:
            res = h.solve(w, want=("z", "resid", "pivots"))
        else:
            res = eng.solve_nodes(b.Qc, b.Rc, b.qd, b.Ac, b.Bc, b.l, b.u, w, z0=z0)
        st = np.asarray(res["status"])
        z = np.asarray(res["z"])
        sel = take[id(b)] if (static and take is not None) else np.ones(len(b), bool)
        if np.any(st[sel] != StatusCode.SUCCESS):
            k = int(np.nonzero((st != StatusCode.SUCCESS) & sel)[0][0])
            bad.append((int(b.dec[k][0]), int(st[k])))
        x_opt[b.dec[sel].ravel()] = z[sel][:, :b.nx].ravel()                         # :440-443
    # ---- multi-node components: pool AVIs, one assemble + one solve per pool shape
    if multis:
        blocks = [_pool_blocks_local(qpn, c, assign) for c in multis]
        sig: Dict[tuple, List[int]] = {}
        for k, bl in enumerate(blocks):
            disjoint = sum(bl["n_i"]) == bl["nd"]
            form = "reference" if (reference_form or not disjoint) else "reduced"
            sig.setdefault((tuple(bl["n_i"]), tuple(bl["m_i"]), tuple(bl["dpos"]), bl["nd"], bl["par"].size, form), []).append(k)
        for key, ks in sorted(sig.items()):
            form = key[-1]
            b0 = blocks[ks[0]]
            p = max(1, b0["par"].size)
            pad = lambda M: np.hstack([M, np.zeros((M.shape[0], p - M.shape[1]))]) if M.shape[1] < p else M
            stack = lambda name: np.stack([colmajor(pad(blocks[k][name]) if name in ("Qp", "Bp") else blocks[k][name]) for k in ks])
            w = np.stack([np.concatenate([x[blocks[k]["par"]], np.zeros(p - blocks[k]["par"].size)]) for k in ks])
            Mc, q, lo, hi, kind = eng.assemble_pools(b0["n_i"], b0["m_i"], b0["dpos"], b0["nd"], stack("Qd"), stack("Qp"),
                                                     np.stack([blocks[k]["qd"] for k in ks]), stack("Ad"), stack("Bp"),
                                                     np.stack([blocks[k]["l"] for k in ks]), np.stack([blocks[k]["u"] for k in ks]),
                                                     w, form=form, share_M=False)
            q = np.asarray(q); Nn = q.shape[-1]
            nd = b0["nd"]
            z0 = np.zeros((len(ks), Nn))
            for t, k in enumerate(ks):
                z0[t, :nd] = x[blocks[k]["dec"]]
                if form == "reference":
                    sm = len(blocks[k]["l"])
                    if sm:                                                          # z0s = [z0; A z0 + B w], :107-108
                        z0[t, Nn - sm:] = blocks[k]["Ad"] @ x[blocks[k]["dec"]] + blocks[k]["Bp"] @ x[blocks[k]["par"]]
            res = eng.solve_avi_batch(np.asarray(Mc).reshape(len(ks), Nn, Nn), q.reshape(len(ks), Nn), np.asarray(lo).reshape(len(ks), Nn),
                                      np.asarray(hi).reshape(len(ks), Nn), z0=z0, kind=np.asarray(kind).reshape(len(ks), Nn))
            st = np.asarray(res["status"]); z = np.asarray(res["z"])
            for t, k in enumerate(ks):
                if int(st[t]) != StatusCode.SUCCESS:
                    bad.append((multis[k], int(st[t])))
                x_opt[blocks[k]["dec"]] = z[t, :nd]
    if bad:
        raise AVISolveError(f"AVI solve error. This might be because one of the qps {list(players)} is unbounded or "
                            f"ill-conditioned. (first failures: {bad[:4]})")
    return x_opt


def _solve_single_reference_form(rec, x, x_opt, eng, bad):
    """One single-node component through the reference's own AVI (xi and slack blocks, convert(): src/avi.jl:113-128) --
    the parity route of solve_qep(reference_form=True)."""
    from .avi import GAVI, StatusCode, solve_gavi
    n, m = rec["dec"].size, len(rec["l"])
    # z = [x_d; xi; lambda], rows [sum xi = 0 ; KKT rows], A = [Ad 0 0]
    M = np.block([[np.zeros((n, n)), np.eye(n), np.zeros((n, m))], [rec["Qd"], np.zeros((n, n)), -rec["Ad"].T]])
    Nn = np.vstack([np.zeros((n, rec["par"].size)), rec["R"]])
    o = np.concatenate([np.zeros(n), rec["qd"]])
    A = np.hstack([rec["Ad"], np.zeros((m, n + m))])
    g = GAVI(M, Nn, o, np.full(2 * n, -INF), np.full(2 * n, INF), A, rec["B"], rec["l"], rec["u"])
    z0 = np.zeros(2 * n + m); z0[:n] = x[rec["dec"]]
    z, status, _ = solve_gavi(g, z0, x[rec["par"]], engine=eng, reference_form=True)
    if status != StatusCode.SUCCESS:
        bad.append((rec["pid"], int(status)))
    x_opt[rec["dec"]] = z[:n]


# ---------------------------------------------------------------------------------------------------------------------
# process_qp for a level: src/algorithm.jl:44-52 over src/qp_processing.jl:151-241
# ---------------------------------------------------------------------------------------------------------------------
_VERIFY_MSGS = {0: "Current point is infeasible when using tolerance {tol}.", 1: "Current point is suboptimal",
                4: "Current point is suboptimal (via QP).", 5: "Solving for duals failed."}


def verify_items(qpn, items, x, engine, tol=1e-4):
    """verify_solution (src/qp_processing.jl:57-149) for MANY (node, child pieces) items: one qpn_verify_nodes call per
    record shape.  items: list of (pid, [child Poly, ...]).  Returns (records, batches, one result dict per item)."""
    eng = engine
    leaf_ids = tuple(pid for pid, ch in items if not ch)
    all_leaf = len(leaf_ids) == len(items)
    if all_leaf:
        recs, batches = _leaf_batches(qpn, list(leaf_ids), eng)
    else:
        recs = [node_record(qpn, pid, ch) for pid, ch in items]
        batches = batch_records(recs)
    out = [None] * len(items)
    for b in batches:
        xd, w = b.gather(x)
        h = b.resident(eng) if all_leaf else None
        if h is not None:
            sol, lam, path = h.verify(xd, w, tol=tol)
        else:
            sol, lam, path = eng.verify_nodes(b.Qc, b.Rc, b.qd, b.Ac, b.Bc, b.l, b.u, xd, w, tol=tol)
        sol = np.asarray(sol); lam = np.asarray(lam); path = np.asarray(path)
        for k, i in enumerate(b.where):
            pth = int(path[k]); ok = bool(sol[k]); mi = int(b.m_true[k])
            out[i] = dict(solution=ok, lam=(lam[k, :mi].copy() if pth in (1, 2, 3, 4) else None),
                          e=None if ok else _VERIFY_MSGS.get(pth, "").format(tol=tol), path=pth)
        b.last = dict(xd=xd, w=w, lam=lam)
    return recs, batches, out


def solution_pieces(qpn, recs, batches, rets, x, engine, want: Sequence[bool], tol=1e-2, max_pieces=64, member_tol=MEMBER_TOL):
    """The solution-graph pieces of MANY nodes around (x, lambda) (process_solution_graph, src/avi.jl:447-477 ->
    comp_indices -> all_Ks, src/avi_solutions.jl:200-215 -> local_piece, :400-496 -> the multipliers eliminated, the
    columns permuted back, :86-87), batched: per record shape one comp_indices pair, one recipes call, one pieces call.
    Returns a list (per item; None where `want` is False) of lists of Poly in global coordinates."""
    from .avi_solutions import _dedupe, _probe_vector
    eng = engine
    x = np.asarray(x, dtype=np.float64)
    out: List[Optional[list]] = [None] * len(recs)
    for b in batches:
        sel = [k for k, i in enumerate(b.where) if want[i]]
        if not sel:
            continue
        n, m, p = b.n, b.m, b.p
        xd, w = b.last["xd"], b.last["w"]
        lam = np.zeros((len(b), m))
        for k, i in enumerate(b.where):
            if want[i]:
                lam[k, :b.m_true[k]] = rets[i]["lam"]
        # the node's own GAVI at z = [x_d; lambda], w = x_p (src/avi.jl:447-477): r1 = Qd x + R w + qd - Ad' lambda, s = Ad x + B w
        r1 = np.einsum("bji,bj->bi", b.Qc, xd) + np.einsum("bji,bj->bi", b.Rc, w) + b.qd - np.einsum("bij,bj->bi", b.Ac, lam)
        s = np.einsum("bji,bj->bi", b.Ac, xd) + np.einsum("bji,bj->bi", b.Bc, w)
        free_lo = np.full((len(sel), n), -INF); free_hi = np.full((len(sel), n), INF)
        m1 = np.asarray(eng.comp_indices(xd[sel], r1[sel], free_lo, free_hi, tol=tol, shift=0))
        m2 = np.asarray(eng.comp_indices(s[sel], lam[sel], b.l[sel], b.u[sel], tol=tol, shift=4)) if m else np.zeros((len(sel), 0), np.uint8)
        for t, k in enumerate(sel):                          # inert padding rows: l = -inf, u = inf, lambda = 0 -> code 6 only
            m2[t, b.m_true[k]:] = 1 << 5
        if m:
            m2 = _refine_row_codes(m2, s[sel], lam[sel], b.l[sel], b.u[sel], CODE_TOL)
        masks = np.concatenate([m1, m2], axis=1).astype(np.uint8)
        ok = ~np.any(masks == 0, axis=1)                     # a zero mask: (x, lambda) is no solution of the GAVI at this tolerance
        pop = np.array([[bin(int(v)).count("1") for v in row] for row in masks], dtype=np.float64)
        total = np.where(ok, np.prod(np.maximum(pop, 1.0), axis=1), 0.0)
        counts = np.minimum(total, max_pieces).astype(np.int64)
        for t, k in enumerate(sel):
            if total[t] > max_pieces:
                warnings.warn(f"node {recs[b.where[k]]['pid']}: {int(total[t])} local recipes, only the first {max_pieces} "
                              "are expanded (max_pieces)")
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        if offsets[-1] == 0:
            for k in sel:
                out[b.where[k]] = []
            continue
        K, node_of = eng.recipes_batch(masks, offsets)
        node_of = np.asarray(node_of)
        rec_of = np.asarray(sel, dtype=np.int32)[node_of]    # recipe -> record inside the batch
        Ar, lr, ur, rows, flags = eng.reduced_pieces(b.Qc, b.Rc, b.qd, b.Ac, b.Bc, b.l, b.u, np.asarray(K), rec_of)
        Ar = np.asarray(Ar); lr = np.asarray(lr); ur = np.asarray(ur); rows = np.asarray(rows); flags = np.asarray(flags)
        for k in sel:
            out[b.where[k]] = []
        colsel = {}                                          # per record: the piece's columns [x_d; x_p present] in ascending order
        for k in sel:
            pk = np.nonzero(b.par[k] >= 0)[0]
            cols = np.concatenate([b.dec[k], b.par[k][pk]])
            order = np.argsort(cols, kind="stable")
            colsel[k] = (cols[order], np.concatenate([np.arange(n), n + pk])[order])
        seen = {b.where[k]: set() for k in sel}
        fallback = {}                                        # per node: the piece the point misses least, for a node none of whose
                                                             # pieces passes (a solution graph is never empty, src/qp_processing.jl:233)
        def admit(i, Pg, miss, key=None):
            # the parent's verify_solution tests feasibility on exactly these normalised rows with 1e-3 (src/qp_processing.jl:86):
            # a piece the point fails here would be "infeasible" there by construction (MEMBER_TOL)
            if miss <= member_tol:
                if key is None:
                    cl, Al_ = Pg.local()
                    key = (cl.tobytes(), (np.round(Al_, 6) + 0.0).tobytes(), np.round(np.concatenate([Pg.l, Pg.u]), 6).tobytes())
                if key not in seen[i]:                       # the reference collects the pieces in a Set (src/avi_solutions.jl:104)
                    seen[i].add(key)
                    out[i].append(Pg)
            elif i not in fallback or miss < fallback[i][0]:
                fallback[i] = (miss, Pg)

        Rmax = Ar.shape[2]
        starts = np.searchsorted(rec_of, np.asarray(sel), side="left"); stops = np.searchsorted(rec_of, np.asarray(sel), side="right")
        for k, t0, t1 in zip(sel, starts.tolist(), stops.tolist()):             # (recipes come record by record, ascending)
            if t1 <= t0:
                continue
            i = b.where[k]
            cols_k, take_k = colsel[k]
            ts = np.arange(t0, t1)
            plain = ts[flags[t0:t1] == 0]
            miss_of = {}
            if plain.size:
                # all of the record's pieces at once: rows over the columns in ascending order, normalised as Poly does
                # (src/sets.jl:76-89: 1e-8 drop, leading coefficient +1), the point's worst violation per piece
                A3 = np.ascontiguousarray(np.swapaxes(Ar[plain][:, take_k, :], 1, 2))        # [pieces, Rmax, columns]
                L2 = lr[plain].astype(np.float64, copy=True); U2 = ur[plain].astype(np.float64, copy=True)
                # (first to unit largest coefficient: the rows come out of the elimination at any scale -- a stationarity row of
                #  size 1e-4 whose 1e-8 entry is dropped, then divided by a leading coefficient of 1e-6, misses its own point by
                #  7e-3, and the parent calls the point infeasible: one pair in 5 000 at n = m = 32 cycled on exactly that)
                big = np.max(np.abs(A3), axis=2)
                sc = np.where(big > 0.0, 1.0 / np.where(big > 0.0, big, 1.0), 1.0)
                A3 *= sc[..., None]
                with np.errstate(invalid="ignore"):
                    L2 = L2 * sc; U2 = U2 * sc
                A3[np.abs(A3) < 1e-8] = 0.0
                nzm = A3 != 0
                has = nzm.any(axis=2)
                lead = np.where(has, np.take_along_axis(A3, nzm.argmax(axis=2)[..., None], axis=2)[..., 0], 1.0)
                nrm = np.abs(lead); neg = has & (lead < 0)
                A3 /= np.where(neg, -nrm, nrm)[..., None]
                with np.errstate(invalid="ignore"):
                    ln, un = L2 / nrm, U2 / nrm
                L2 = np.where(neg, -un, ln); U2 = np.where(neg, -ln, un)
                ax = A3 @ x[cols_k]
                live = has & (np.arange(Rmax)[None, :] < rows[plain][:, None])
                viol = np.where(live, np.maximum(L2 - ax, ax - U2), 0.0)
                worst = np.max(viol, axis=1, initial=0.0)
                # _dedupe's own quick test (equal normals project equally on a random vector; all-zero rows) for all pieces at once:
                # the few that have something to merge go through it, the others are taken as they are
                valid = np.arange(Rmax)[None, :] < rows[plain][:, None]
                with np.errstate(invalid="ignore"):
                    hs = np.sort(np.where(valid, A3 @ _probe_vector(A3.shape[2]), np.inf), axis=1)
                    close = np.diff(hs, axis=1) <= 1e-7 * (1.0 + np.abs(hs[:, 1:]))
                merge = np.any(close & np.isfinite(hs[:, 1:]), axis=1) | np.any(valid & ~has, axis=1)
                A6 = np.round(A3, 6) + 0.0                      # (the keys of the node's piece SET, rounded once for all pieces)
                LU6 = np.round(np.concatenate([L2, U2], axis=1), 6)
                ckey = cols_k.tobytes()
                for j, t in enumerate(plain.tolist()):
                    r = int(rows[t])
                    key = None if merge[j] else (ckey, A6[j, :r].tobytes(), np.concatenate([LU6[j, :r], LU6[j, Rmax:Rmax + r]]).tobytes())
                    miss_of[t] = (float(worst[j]), A3[j, :r].copy(), L2[j, :r].copy(), U2[j, :r].copy(), bool(merge[j]), key)
            for t in ts.tolist():
                if flags[t]:
                    P = _reduce_on_host(b, k, np.asarray(K)[t], eng)
                    if P is None:
                        continue
                    Ah = np.ascontiguousarray(P[0][:, take_k]); big = np.max(np.abs(Ah), axis=1, initial=0.0)
                    sc = np.where(big > 0.0, 1.0 / np.where(big > 0.0, big, 1.0), 1.0)
                    with np.errstate(invalid="ignore"):
                        Pg = _dedupe(Poly.from_sorted(qpn.num_vars, cols_k, Ah * sc[:, None], P[1] * sc, P[2] * sc))
                    cl, Al_ = Pg.local()
                    axp = Al_ @ x[cl]
                    admit(i, Pg, float(np.max(np.maximum(Pg.l - axp, axp - Pg.u), initial=0.0)))
                else:
                    miss, Aj, lj, uj, mg, key = miss_of[t]
                    Pg = Poly.from_sorted(qpn.num_vars, cols_k, Aj, lj, uj, normalise=False)
                    admit(i, _dedupe(Pg) if mg else Pg, miss, key)
        for i, (miss, Pg) in fallback.items():
            if not out[i]:
                out[i].append(Pg)
    return out


def _refine_row_codes(m2, s, lam, l, u, mt):
    """Of the codes comp_indices allows on a constraint row (tolerance 1e-2), keep those whose OWN condition the point meets
    within `mt`: code 5 (s = l, lambda >= 0), 6 (lambda = 0, l <= s <= u), 7 (s = u, lambda <= 0), 8 (l = s = u).  A recipe's piece
    contains the point exactly when every row's condition holds (the lifted piece is a product of per-row conditions, src/
    avi_solutions.jl:413-432), so after this every enumerated recipe's piece contains the point -- see MEMBER_TOL.  A row left
    without any code keeps the allowed code it violates least."""
    viol = np.full(m2.shape + (4,), np.inf)
    with np.errstate(invalid="ignore"):
        viol[..., 0] = np.maximum(np.abs(s - l), np.maximum(-lam, 0.0))
        viol[..., 1] = np.maximum(np.abs(lam), np.maximum(np.maximum(l - s, s - u), 0.0))
        viol[..., 2] = np.maximum(np.abs(s - u), np.maximum(lam, 0.0))
        viol[..., 3] = np.maximum(np.abs(s - l), np.abs(s - u))
    viol = np.where(np.isnan(viol), np.inf, viol)
    bits = (m2[..., None] >> np.arange(4, 8)) & 1
    viol = np.where(bits == 1, viol, np.inf)
    keep = viol <= mt
    best = np.argmin(viol, axis=-1)
    none = ~keep.any(axis=-1) & (m2 != 0)
    keep[none, best[none]] = True
    out = np.zeros_like(m2)
    for c in range(4):
        out |= (keep[..., c].astype(np.uint8) << (4 + c)).astype(np.uint8)
    return out


def _reduce_on_host(b, k, Krow, eng):
    """A piece the device elimination flagged (a multiplier no equality row pins: Fourier-Motzkin, polyhedral) goes
    through the host restatement (avi_solutions.eliminate_multipliers)."""
    from .avi_solutions import eliminate_multipliers
    Ap, lp, up, keep = eng.local_pieces(b.Qc[k:k + 1], b.Rc[k:k + 1], b.qd[k:k + 1], b.Ac[k:k + 1], b.Bc[k:k + 1], b.l[k:k + 1],
                                        b.u[k:k + 1], np.asarray(Krow, dtype=np.uint8)[None], node_of=np.zeros(1, np.int32))
    rows = np.asarray(keep[0]).astype(bool)
    P = Poly(np.asarray(Ap[0]).T[rows], np.asarray(lp[0])[rows], np.asarray(up[0])[rows], normalise=False)
    try:
        Pl = eliminate_multipliers(P, b.n, b.m)
    except RuntimeError:
        return None
    return Pl.vectorize()


def process_level(qpn, players: Sequence[int], x, S: Dict[int, list], engine=None, exploration_vertices=0):
    """results = map(process_qp(qpn, id, x, S) for id in players) (src/algorithm.jl:44-52) as batches: every node under
    every combination of its children's pieces (src/qp_processing.jl:162-205) in one verify call per record shape, the
    solution graphs of the optimal ones in one call chain per record shape.  Returns one process_qp result per player."""
    from .avi import _eng
    from .qp_processing import combine_many
    eng = _eng(engine)
    x = np.asarray(x, dtype=np.float64)
    # process_qp is a function of x on the node's subtree columns and of its children's solution graphs: a node for which
    # neither has changed since the last sweep gets the result it got then (the reference recomputes it, src/algorithm.jl:47 --
    # to the same value).  On a net of many independent clusters most nodes are at rest after the first sweeps.
    memo = qpn.__dict__.setdefault("_process_memo", {})
    all_players = list(players)
    xkey = {pid: x[_subtree_cols(qpn, pid)].tobytes() for pid in all_players}
    hits = {}
    for pid in all_players:
        ent = memo.get(pid)
        if ent is not None and ent["xkey"] == xkey[pid] and ent["ev"] == exploration_vertices and \
                all(S.get(j) is g for j, g in ent["kids"].items()):
            hits[pid] = ent["result"]
    all_leaf_level = all(not qpn.network_edges[pid] for pid in all_players)
    if not all_leaf_level:
        players = [pid for pid in all_players if pid not in hits]      # (a level of childless nodes keeps its one resident batch:
                                                                       #  all are verified in one launch, hits skip the pieces)
    items, owner = [], []
    combos_of = {}
    for pid in players:
        children = sorted(qpn.network_edges[pid])
        if children:
            cards = [range(len(S[j])) for j in children]
            if any(len(c) < 1 for c in cards):
                raise RuntimeError("Solution graphs were not properly populated.")
            combos = list(itertools.product(*cards))
        else:
            combos = [()]
        combos_of[pid] = (children, combos)
        for combo in combos:
            items.append((pid, [S[j][ji] for j, ji in zip(children, combo)]))
            owner.append(pid)
    recs, batches, rets = verify_items(qpn, items, x, eng)
    results = {}
    first = {}
    for i, (pid, _) in enumerate(items):
        first.setdefault(pid, i)
    want = [False] * len(items)
    for pid in players:
        children, combos = combos_of[pid]
        i0 = first[pid]
        fail = next((t for t in range(len(combos)) if not rets[i0 + t]["solution"]), None)
        if fail is not None:                                                        # the first failure in product order, :206-216
            results[pid] = dict(solution=False, e=rets[i0 + fail]["e"], failed=False,
                                subpiece_assignments={j: ji for j, ji in zip(children, combos[fail])} if children else {})
            continue
        gen = (pid not in qpn.network_depth_map[1]) or qpn.options.gen_solution_map
        results[pid] = dict(solution=True, S=None, failed=False)
        if pid in hits:
            continue
        if gen:
            for t in range(len(combos)):
                want[i0 + t] = True
    if any(want):
        pieces = solution_pieces(qpn, recs, batches, rets, x, eng, want)
        jobs, job_pid = [], []
        for pid in players:
            children, combos = combos_of[pid]
            i0 = first[pid]
            if not want[i0]:
                continue
            per_combo = [pieces[i0 + t] for t in range(len(combos))]
            if len(combos) == 1:
                results[pid]["S"] = per_combo[0]                                    # combine(...) with one solution set, :271-272
            else:
                jobs.append(([[S[j][ji] for j, ji in zip(children, combo)] for combo in combos], per_combo))
                job_pid.append(pid)
        if jobs:                                                                    # all kinks of the level: one batch of LPs
            for pid, got in zip(job_pid, combine_many(jobs, x, eng)):
                if isinstance(got, RuntimeError):
                    results[pid] = dict(solution=False, failed=True, S=None)        # :219-223
                else:
                    results[pid]["S"] = got
        for pid in players:
            if results[pid].get("solution") and want[first[pid]] and len(results[pid]["S"]) == 0:
                raise RuntimeError("This shouldn't happen. Solution graph is empty.")
    for pid in all_players:
        if pid in hits:
            results[pid] = hits[pid]
        else:
            memo[pid] = dict(xkey=xkey[pid], ev=exploration_vertices, result=results[pid],
                             kids={j: S.get(j) for j in qpn.network_edges[pid]})
    return [results[pid] for pid in all_players]
