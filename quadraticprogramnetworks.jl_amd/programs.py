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


@dataclass
class Quadratic:
    """f(x) = 1/2 x'Qx + q'x + k over ALL variables of the net (src/programs.jl:16-24)."""
    Q: np.ndarray
    q: np.ndarray
    k: float = 0.0

    def __call__(self, x):
        return 0.5 * x @ (self.Q @ x) + x @ self.q + self.k


class Poly:
    """Closed polyhedron {x : l <= A x <= u} as normalised rows (src/sets.jl:68-92, :151-154).

    Every row is scaled so that its leading non-zero coefficient is +1 (lexico-positive); a
    negative leading coefficient flips the row and swaps/negates its bounds (:83-88).  The
    reference keeps rows in a Set (iteration order = hash order, :123-125, :213-221); here rows
    keep the order they were given, so row indices are reproducible.
    """

    def __init__(self, A, l, u, normalise=True, tol=1e-8, open_lo=None, open_hi=None):
        A = np.atleast_2d(np.asarray(A, dtype=np.float64)).copy()
        l = np.asarray(l, dtype=np.float64).copy()
        u = np.asarray(u, dtype=np.float64).copy()
        # relations of the bounds (src/sets.jl:68-92: rl, ru in {<=, <}): True = strict.  Closed by default.
        olo = np.zeros(l.shape, bool) if open_lo is None else np.asarray(open_lo, bool).copy()
        ohi = np.zeros(u.shape, bool) if open_hi is None else np.asarray(open_hi, bool).copy()
        if A.size == 0:
            A = A.reshape(0, A.shape[-1] if A.ndim == 2 else 0)
        assert A.shape[0] == l.shape[0] == u.shape[0]
        if normalise:
            for i in range(A.shape[0]):
                a = A[i]
                a[np.abs(a) < tol] = 0.0                      # droptol!, :76
                nz = np.nonzero(a)[0]
                if nz.size == 0:
                    continue
                lead = a[nz[0]]
                nrm = abs(lead)
                if lead >= 0:
                    A[i] = a / nrm; l[i] /= nrm; u[i] /= nrm
                else:
                    A[i] = -a / nrm
                    l[i], u[i] = -u[i] / nrm, -l[i] / nrm
                    olo[i], ohi[i] = ohi[i], olo[i]            # the relations swap with the bounds (:88)
        self.A, self.l, self.u = A, l, u
        self.open_lo, self.open_hi = olo, ohi

    def __len__(self):
        return self.A.shape[0]

    def vectorize(self):
        """(A, l, u), src/sets.jl:213-221."""
        return self.A, self.l, self.u

    def open_bounds(self):
        """(open_low, open_hi), src/sets.jl:354-356: strict relations on FINITE bounds."""
        return self.open_lo & np.isfinite(self.l), self.open_hi & np.isfinite(self.u)

    def contains(self, x, tol=1e-6):
        ax = self.A @ x                                                           # src/sets.jl:850-853: rl(l - tol, ax) && ru(ax - tol, u)
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
    def add_constraint(self, A, l, u) -> int:
        A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        assert A.shape[1] == self.num_vars
        cid = max(self.constraints.keys(), default=0) + 1
        self.constraints[cid] = Constraint(Poly(A, l, u))
        return cid

    # -- src/programs.jl:172-201
    def add_qp(self, Q, q, con_inds, var_indices, k=0.0) -> int:
        Q = np.asarray(Q, dtype=np.float64)
        assert Q.shape == (self.num_vars, self.num_vars)
        pid = max(self.qps.keys(), default=0) + 1
        self.qps[pid] = QP(Quadratic(Q, np.asarray(q, dtype=np.float64), float(k)),
                           list(con_inds), [int(v) for v in var_indices])
        return pid

    # -- src/programs.jl:214-285
    def add_edges(self, edge_list):
        N = len(self.qps)
        A = np.zeros((N, N), dtype=bool)
        for (i, j) in edge_list:
            if i == j:
                raise ValueError(f"Cannot have self edges. (In this case, node {i} -> {i}).")
            A[i - 1, j - 1] = True
        R = np.zeros((N, N), dtype=bool)
        An = A.copy()
        for n in range(2, N + 1):
            R |= An
            An = (An.astype(int) @ A.astype(int)) > 0
            for i in range(N):
                if An[i, i]:
                    raise ValueError(f"Cycle detected. (node {i + 1} -> {i + 1} after {n} transitions.)")
                for j in range(N):
                    if A[i, j] and An[i, j]:
                        A[i, j] = False                      # redundant edge (transitive reduction)
        R |= An if N > 1 else A
        R |= A
        # depth map, :249-269
        depth_map: Dict[int, Set[int]] = {}
        deleted: Set[int] = set()
        d = 0
        Rd = R.copy()
        rows = list(range(N))
        while len(deleted) < N:
            at_depth = {i for i in range(N) if not Rd[:, i].any()} - deleted
            if not at_depth:
                raise ValueError("Something appears wrong with the graph structure.")
            d += 1
            depth_map[d] = {i + 1 for i in at_depth}
            deleted |= at_depth
            rows = [i for i in range(N) if i not in deleted]
            Rd = R[rows, :] if rows else np.zeros((0, N), dtype=bool)
        self.network_depth_map = depth_map
        for i in range(N):
            self.network_edges[i + 1] = {j + 1 for j in range(N) if A[i, j]}
            self.reachable_nodes[i + 1] = {j + 1 for j in range(N) if R[i, j]}

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
        inds = set(self.qps[pid].var_indices)
        for j in self.reachable_nodes[pid]:
            inds |= set(self.qps[j].var_indices)
        return sorted(inds)
