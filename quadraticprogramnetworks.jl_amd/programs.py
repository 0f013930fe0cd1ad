"""QPNet data model: the INPUT CONTRACT of the hot path (mirror of src/programs.jl, numeric only).

The reference builds Q, q, A, l, u from Symbolics expressions (src/programs.jl:147-201); that
front-end runs once per problem and is out of scope.  Here the same records are given numerically.
Names follow the reference (QPNet, QP, Quadratic, Constraint, QPNetOptions, add_constraint!,
add_qp!, add_edges!, assign_constraint_groups!, set_options!, decision_inds, num_levels).
Indices are 0-based (Julia: 1-based); node and constraint ids start at 1 as in the reference.
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Set

import numpy as np

INF = np.inf


DENSE_LIMIT = 1 << 26          # entries: the dense global views (.Q, .A) refuse to materialise beyond this


def _positions(idx, wanted):
    """For sorted unique `idx`: position of every entry of `wanted` in idx, and a mask of the ones present."""
    wanted = np.asarray(wanted, dtype=np.int64)
    if idx.size == 0:
        return np.zeros(wanted.shape, np.int64), np.zeros(wanted.shape, bool)
    pos = np.searchsorted(idx, wanted)
    pos = np.minimum(pos, idx.size - 1)
    return pos, idx[pos] == wanted


class Quadratic:
    """f(x) = 1/2 x'Qx + q'x + k over ALL variables of the net (src/programs.jl:16-24).

    The reference keeps Q as an n_total x n_total SparseMatrixCSC per QP.  Here Q is held either as the dense array it was
    given as, or in LOCAL form -- the sorted indices `idx` of the variables the cost touches and the dense |idx| x |idx| block
    on them (`from_local`; what a 10 000-node net needs: a dense n_total^2 array per node does not exist there).  `.Q` / `.q`
    are the dense global views (small nets, tests); the level batches read blocks through `block` / `q_at` / `row_support`."""

    def __init__(self, Q, q, k: float = 0.0):
        if hasattr(Q, "tocoo"):                                   # scipy.sparse: straight to the local form
            coo = Q.tocoo()
            qv = np.asarray(q, dtype=np.float64)
            idx = np.unique(np.concatenate([coo.row[coo.data != 0], coo.col[coo.data != 0], np.nonzero(qv)[0]])).astype(np.int64)
            pr, _ = _positions(idx, coo.row); pc, _ = _positions(idx, coo.col)
            Ql = np.zeros((idx.size, idx.size)); np.add.at(Ql, (pr, pc), coo.data)
            self._nv, self._idx, self._Ql, self._ql = int(Q.shape[0]), idx, Ql, qv[idx].copy()
            self._Q = self._q = None
        else:
            self._Q = np.asarray(Q, dtype=np.float64)
            self._q = np.asarray(q, dtype=np.float64)
            self._nv = int(self._Q.shape[0])
            self._idx = self._Ql = self._ql = None
        self.k = float(k)

    @classmethod
    def from_local(cls, nv: int, idx, Qloc, qloc, k: float = 0.0) -> "Quadratic":
        f = cls.__new__(cls)
        idx = np.asarray(idx, dtype=np.int64)
        order = np.argsort(idx, kind="stable")
        Ql = np.asarray(Qloc, dtype=np.float64)[np.ix_(order, order)]
        f._nv, f._idx, f._Ql, f._ql = int(nv), idx[order], Ql, np.asarray(qloc, dtype=np.float64)[order]
        if np.any(np.diff(f._idx) == 0):
            raise ValueError("Quadratic.from_local: repeated variable index")
        f._Q = f._q = None
        f.k = float(k)
        return f

    @property
    def num_vars(self) -> int:
        return self._nv

    def local(self):
        """(idx, Qloc, qloc): the variables the cost touches (sorted) and the blocks on them."""
        if self._idx is None:
            nz = (self._Q != 0)
            idx = np.nonzero(nz.any(axis=0) | nz.any(axis=1) | (self._q != 0))[0].astype(np.int64)
            self._idx, self._Ql, self._ql = idx, self._Q[np.ix_(idx, idx)], self._q[idx]
        return self._idx, self._Ql, self._ql

    @property
    def Q(self):
        if self._Q is None:
            if self._nv * self._nv > DENSE_LIMIT:
                raise MemoryError(f"Quadratic.Q: a dense {self._nv} x {self._nv} view is refused; use block()/local()")
            Q = np.zeros((self._nv, self._nv)); Q[np.ix_(self._idx, self._idx)] = self._Ql
            self._Q = Q
        return self._Q

    @property
    def q(self):
        if self._q is None:
            q = np.zeros(self._nv); q[self._idx] = self._ql
            self._q = q
        return self._q

    def block(self, rows, cols):
        """Dense Q[rows, cols] for global index lists."""
        idx, Ql, _ = self.local()
        pr, okr = _positions(idx, rows); pc, okc = _positions(idx, cols)
        out = Ql[np.ix_(pr, pc)]
        out = np.where(okr[:, None] & okc[None, :], out, 0.0)
        return out

    def q_at(self, rows):
        idx, _, ql = self.local()
        pr, ok = _positions(idx, rows)
        return np.where(ok, ql[pr] if idx.size else 0.0, 0.0)

    def row_support(self, rows):
        """Sorted global columns j with Q[rows, j] != 0 for some row."""
        idx, Ql, _ = self.local()
        pr, ok = _positions(idx, rows)
        if not ok.any():
            return np.zeros(0, np.int64)
        return idx[(Ql[pr[ok]] != 0).any(axis=0)]

    def __call__(self, x):
        idx, Ql, ql = self.local()
        xl = np.asarray(x, dtype=np.float64)[idx]
        return 0.5 * xl @ (Ql @ xl) + xl @ ql + self.k


class Poly:
    """Closed polyhedron {x : l <= A x <= u} as normalised rows (src/sets.jl:68-92, :151-154).

    Every row is scaled so that its leading non-zero coefficient is +1 (lexico-positive); a
    negative leading coefficient flips the row and swaps/negates its bounds (:83-88).  The
    reference keeps rows in a Set (iteration order = hash order, :123-125, :213-221); here rows
    keep the order they were given, so row indices are reproducible.

    Rows are held densely over all variables (`Poly(A, l, u)`) or in LOCAL form over the sorted columns `cols` they touch
    (`Poly.from_local`; the solution-graph pieces of a large net).  `.A` is the dense global view; `local()` / `block()` serve
    the level batches.
    """

    def __init__(self, A, l, u, normalise=True, tol=1e-8, open_lo=None, open_hi=None):
        A = np.atleast_2d(np.asarray(A, dtype=np.float64)).copy()
        if A.size == 0:
            A = A.reshape(0, A.shape[-1] if A.ndim == 2 else 0)
        self._init(A.shape[1], None, A, l, u, normalise, tol, open_lo, open_hi)

    @classmethod
    def from_local(cls, ncols: int, cols, A, l, u, normalise=True, tol=1e-8, open_lo=None, open_hi=None) -> "Poly":
        """Rows over the variables `cols` only (any order, no repeats): A is [rows, len(cols)]."""
        P = cls.__new__(cls)
        cols = np.asarray(cols, dtype=np.int64)
        order = np.argsort(cols, kind="stable")
        A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        A = A.reshape(-1, cols.size)[:, order].copy()
        if np.any(np.diff(cols[order]) == 0):
            raise ValueError("Poly.from_local: repeated column index")
        P._init(int(ncols), cols[order], A, l, u, normalise, tol, open_lo, open_hi)
        return P

    @classmethod
    def from_sorted(cls, ncols: int, cols, A, l, u, normalise=True) -> "Poly":
        """from_local for columns already ascending and without repeats (the caller vouches: the level batches, whose column
        lists are a record's own sorted index sets); A [rows, len(cols)] is taken over, not copied.  normalise=False: the rows
        are normalised already (level_batch does that for all pieces of a record at once)."""
        P = cls.__new__(cls)
        P._init(int(ncols), cols, A, l, u, normalise, 1e-8, None, None)
        return P

    def _init(self, ncols, cols, A, l, u, normalise, tol, open_lo, open_hi):
        l = np.asarray(l, dtype=np.float64).copy()
        u = np.asarray(u, dtype=np.float64).copy()
        # relations of the bounds (src/sets.jl:68-92: rl, ru in {<=, <}): True = strict.  Closed by default.
        olo = np.zeros(l.shape, bool) if open_lo is None else np.asarray(open_lo, bool).copy()
        ohi = np.zeros(u.shape, bool) if open_hi is None else np.asarray(open_hi, bool).copy()
        assert A.shape[0] == l.shape[0] == u.shape[0]
        if normalise and A.shape[0]:
            A[np.abs(A) < tol] = 0.0                              # droptol!, :76
            nzm = A != 0
            has = nzm.any(axis=1)
            first = np.argmax(nzm, axis=1)                        # leading non-zero (columns ascend with the variable index)
            lead = np.where(has, A[np.arange(A.shape[0]), first], 1.0)
            nrm = np.abs(lead)
            neg = has & (lead < 0)
            A /= np.where(neg, -nrm, nrm)[:, None]
            ln, un = l / nrm, u / nrm
            l = np.where(neg, -un, ln); u = np.where(neg, -ln, un)
            olo, ohi = np.where(neg, ohi, olo), np.where(neg, olo, ohi)   # the relations swap with the bounds (:88)
        self.ncols = int(ncols)
        self._cols = cols                                         # None: A is dense over all variables
        self._A = A
        self.l, self.u = l, u
        self.open_lo, self.open_hi = olo, ohi

    @property
    def A(self):
        if self._cols is None:
            return self._A
        if self._A.shape[0] * self.ncols > DENSE_LIMIT:
            raise MemoryError(f"Poly.A: a dense {self._A.shape[0]} x {self.ncols} view is refused; use local()/block()")
        A = np.zeros((self._A.shape[0], self.ncols)); A[:, self._cols] = self._A
        return A

    def local(self):
        """(cols, A_local): the sorted variables the rows touch and the coefficients on them."""
        if self._cols is None:
            cols = np.nonzero((self._A != 0).any(axis=0))[0].astype(np.int64)
            return cols, self._A[:, cols]
        return self._cols, self._A

    def block(self, cols):
        """Dense A[:, cols] for a global index list."""
        cols = np.asarray(cols, dtype=np.int64)
        if self._cols is None:
            return self._A[:, cols]
        if cols.size == self._cols.size and np.array_equal(cols, self._cols):
            return self._A                                         # (the rows' own columns: no copy -- callers read it or stack it)
        pos, ok = _positions(self._cols, cols)
        return np.where(ok[None, :], self._A[:, pos] if self._cols.size else 0.0, 0.0)

    def support(self):
        return self.local()[0]

    def __len__(self):
        return self._A.shape[0]

    def vectorize(self):
        """(A, l, u), src/sets.jl:213-221."""
        return self.A, self.l, self.u

    def open_bounds(self):
        """(open_low, open_hi), src/sets.jl:354-356: strict relations on FINITE bounds."""
        return self.open_lo & np.isfinite(self.l), self.open_hi & np.isfinite(self.u)

    def contains(self, x, tol=1e-6):
        cols, Al = self.local()
        ax = Al @ np.asarray(x, dtype=np.float64)[cols]                           # src/sets.jl:850-853: rl(l - tol, ax) && ru(ax - tol, u)
        lo_ok = np.where(self.open_lo, self.l - tol < ax, self.l - tol <= ax)
        hi_ok = np.where(self.open_hi, ax - tol < self.u, ax - tol <= self.u)
        return bool(np.all(lo_ok) and np.all(hi_ok))


@dataclass
class Constraint:
    poly: Poly
    group_mapping: Dict[int, int] = field(default_factory=dict)


@dataclass
class QP:
    f: Quadratic
    constraint_indices: List[int]
    var_indices: List[int]


@dataclass
class QPNetOptions:
    """src/programs.jl:61-77 -- same field names and defaults."""
    shared_variable_mode: str = "SHARED_DUAL"
    max_iters: int = 150
    tol: float = 1e-4
    high_dimension: bool = False
    high_dimension_max_iters: int = 10
    num_projections: int = 4
    make_requests: bool = False
    exploration_vertices: int = 0
    try_hull: bool = False
    debug_visualize: bool = False
    gen_solution_map: bool = False
    levels_to_remove_subsets: object = None      # NaturalNumbers() in the reference
    check_convexity: bool = False
    check_for_cycling: bool = True
    perturb_to_continue: bool = True


class QPNet:
    """src/programs.jl:79-116."""

    def __init__(self, num_vars: int):
        self.num_vars = int(num_vars)
        self.qps: Dict[int, QP] = {}
        self.constraints: Dict[int, Constraint] = {}
        self.network_edges: Dict[int, Set[int]] = {}
        self.reachable_nodes: Dict[int, Set[int]] = {}
        self.network_depth_map: Dict[int, Set[int]] = {}
        self.options = QPNetOptions()
        self.problem_data: dict = {}
        self.iterate_cache: Dict[int, list] = {}
        self.default_initialization = np.zeros(self.num_vars)

    # -- src/programs.jl:147-170
    def add_constraint(self, A, l, u, cols=None) -> int:
        """Rows l <= A x <= u.  `cols` given: A is [rows, len(cols)] over those variables only (local form)."""
        cid = max(self.constraints.keys(), default=0) + 1
        if cols is not None:
            self.constraints[cid] = Constraint(Poly.from_local(self.num_vars, cols, A, l, u))
            return cid
        A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        assert A.shape[1] == self.num_vars
        self.constraints[cid] = Constraint(Poly(A, l, u))
        return cid

    # -- src/programs.jl:172-201
    def add_qp(self, Q, q, con_inds, var_indices, k=0.0, idx=None) -> int:
        """Q, q over all variables (dense or scipy.sparse) -- or, with `idx`, the blocks on the variables idx only."""
        pid = max(self.qps.keys(), default=0) + 1
        if idx is not None:
            f = Quadratic.from_local(self.num_vars, idx, Q, q, float(k))
        else:
            assert tuple(Q.shape) == (self.num_vars, self.num_vars)
            f = Quadratic(Q, q, float(k))
        self.qps[pid] = QP(f, list(con_inds), [int(v) for v in var_indices])
        return pid

    # -- src/programs.jl:214-285
    def add_edges(self, edge_list):
        """Transitive reduction + cycle check (create_minimal_adj_matrix, :214-242), reachability, depth map (:249-269).
        The reference does it with powers of the dense adjacency matrix; the same sets come out of one pass in
        topological order over adjacency lists, which a 10 000-node net needs."""
        N = len(self.qps)
        kids: Dict[int, Set[int]] = {i: set() for i in range(1, N + 1)}
        for (i, j) in edge_list:
            if i == j:
                raise ValueError(f"Cannot have self edges. (In this case, node {i} -> {i}).")
            kids[int(i)].add(int(j))
        indeg = {i: 0 for i in kids}
        for i, js in kids.items():
            for j in js:
                indeg[j] += 1
        order, ready = [], sorted(i for i, d in indeg.items() if d == 0)
        left = dict(indeg)
        while ready:
            i = ready.pop()
            order.append(i)
            for j in kids[i]:
                left[j] -= 1
                if left[j] == 0:
                    ready.append(j)
        if len(order) < N:
            bad = min(i for i, d in left.items() if d > 0)
            raise ValueError(f"Cycle detected. (node {bad} -> {bad} after some transitions.)")
        reach: Dict[int, Set[int]] = {}
        for i in reversed(order):                                # children before parents
            r = set(kids[i])
            for j in kids[i]:
                r |= reach[j]
            reach[i] = r
        for i in kids:                                           # an edge i -> j is redundant when another child reaches j
            kids[i] = {j for j in kids[i] if not any(j in reach[k] for k in kids[i] if k != j)}
        depth = {}
        parents: Dict[int, List[int]] = {i: [] for i in kids}
        for i, js in reach.items():                              # depth = 1 + the longest chain of ancestors (:249-269)
            for j in js:
                parents[j].append(i)
        for i in order:
            depth[i] = 1 + max((depth[a] for a in parents[i]), default=0)
        depth_map: Dict[int, Set[int]] = {}
        for i, d in depth.items():
            depth_map.setdefault(d, set()).add(i)
        self.network_depth_map = dict(sorted(depth_map.items()))
        self.network_edges = kids
        self.reachable_nodes = reach
        self._dec_cache = {}

    # -- src/programs.jl:293-310
    def assign_constraint_groups(self, group_map=None):
        group_map = group_map or {}
        for cid, con in self.constraints.items():
            for pid, qp in self.qps.items():
                if cid in qp.constraint_indices:
                    if cid in group_map:
                        if pid not in group_map[cid]:
                            raise ValueError(f"group map for constraint {cid} lacks player {pid}")
                        con.group_mapping[pid] = group_map[cid][pid]
                    else:
                        con.group_mapping[pid] = pid

    # -- src/programs.jl:312-320: unknown names only warn
    def set_options(self, **kwargs):
        for k, v in kwargs.items():
            if hasattr(self.options, k):
                setattr(self.options, k, v)
            else:
                warnings.warn(f"Invalid option name {k} with value {v}, skipping")

    def num_levels(self) -> int:
        return len(self.network_depth_map)

    # -- src/programs.jl:340-346
    def decision_inds(self, pid: int) -> List[int]:
        cache = self.__dict__.setdefault("_dec_cache", {})
        got = cache.get(pid)
        if got is None:
            inds = set(self.qps[pid].var_indices)
            for j in self.reachable_nodes[pid]:
                inds |= set(self.qps[j].var_indices)
            got = cache[pid] = sorted(inds)
        return list(got)
