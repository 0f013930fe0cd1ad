"""Analytic known answers through the whole host loop (solve -> solve_base -> process_qp / verify_solution -> solve_qep ->
pool assembly -> AVI solve): two-player zero-sum matrix games as single-level QPNets -- each player a QP over its mixed strategy
(simplex: one equality row + non-negativity), bilinear cost x'Ay / -x'Ay, i.e. degenerate (Q_dd = 0) nodes whose pool AVI is
monotone but not strictly so.  The mixed equilibria are textbook: rock-paper-scissors (1/3, 1/3, 1/3), matching pennies (1/2, 1/2),
the 2 x 2 game [[a, b], [c, d]] without a saddle point p = (d - c) / (a - b - c + d), q = (d - b) / (a - b - c + d).
And quantity competition with linear demand (price a - b sum q, unit cost c; player i: min  b q_i^2 + b q_i sum_{j != i} q_j - (a - c) q_i,
q_i >= 0) in its classical variants, on one, two and three levels of the network: Cournot-Nash with n firms q_i = (a - c) / ((n + 1) b);
Stackelberg leader-follower (a - c) / (2b), (a - c) / (4b); a three-level chain (a - c) / (2b), / (4b), / (8b); a leader above
two Nash followers (a - c) / (2b), then (a - c) / (6b) each.
And the reference's own `four_player_matrix_game` example (examples/four_player_matrix_game.jl) under ten edge lists -- one to four
levels, several players per level, a child with two parents -- against `backward_substitution`: the equilibrium of quadratic players
whose constraints stay inactive, by elimination of the descendants' affine responses (linear solves only).
Not reference-held vectors -- the reference has none for such nets -- but independent of every restatement in this repository."""
import numpy as np
import pytest

import qpn_amd  # noqa: F401
from qpn_amd import algorithm
from qpn_amd.programs import QPNet

INF = np.inf


def matrix_game(A, reg=0.0):
    r, c = A.shape
    nv = r + c
    net = QPNet(nv)
    c1 = net.add_constraint(np.hstack([np.eye(r), np.zeros((r, c))]), np.zeros(r), np.full(r, INF))
    s1 = net.add_constraint(np.hstack([np.ones((1, r)), np.zeros((1, c))]), [1.0], [1.0])
    c2 = net.add_constraint(np.hstack([np.zeros((c, r)), np.eye(c)]), np.zeros(c), np.full(c, INF))
    s2 = net.add_constraint(np.hstack([np.zeros((1, r)), np.ones((1, c))]), [1.0], [1.0])
    Q1 = np.zeros((nv, nv)); Q1[:r, r:] = A; Q1[r:, :r] = A.T; Q1[:r, :r] += reg * np.eye(r)
    Q2 = np.zeros((nv, nv)); Q2[:r, r:] = -A; Q2[r:, :r] = -A.T; Q2[r:, r:] += reg * np.eye(c)
    net.add_qp(Q1, np.zeros(nv), [c1, s1], list(range(r)), 0.0)
    net.add_qp(Q2, np.zeros(nv), [c2, s2], list(range(r, nv)), 0.0)
    net.add_edges([])
    net.assign_constraint_groups()
    net.set_options(debug_visualize=False)
    net.default_initialization = np.concatenate([np.eye(r)[0], np.eye(c)[0]])      # a pure-strategy start: not an equilibrium
    return net


RPS = np.array([[0.0, -1, 1], [1, 0, -1], [-1, 1, 0]])
GAMES = [("rock-paper-scissors", RPS, 0.0, np.full(6, 1 / 3)),
         ("matching pennies", np.array([[1.0, -1], [-1, 1]]), 0.0, np.full(4, 0.5)),
         ("2 x 2 without a saddle point", np.array([[2.0, -1], [-1, 1]]), 0.0, np.array([0.4, 0.6, 0.4, 0.6])),
         ("rock-paper-scissors, regularised (strictly monotone)", RPS, 0.5, np.full(6, 1 / 3)),
         # (the row player MINIMISES x'Ay, the column player maximises it: row 0 dominates, then column 0)
         ("a saddle point in pure strategies", np.array([[3.0, 1], [4, 2]]), 0.0, np.array([1.0, 0.0, 1.0, 0.0]))]


def quantity_game(n, a, b, c, edges=()):
    """n firms, variable i = firm i's quantity; edges (i, j): firm i leads firm j (i one level above j)."""
    net = QPNet(n)
    pids = []
    for i in range(n):
        cid = net.add_constraint(np.eye(n)[i][None], [0.0], [INF])
        Q = np.zeros((n, n)); Q[i, i] = 2 * b
        for j in range(n):
            if j != i:
                Q[i, j] = Q[j, i] = b
        lin = np.zeros(n); lin[i] = -(a - c)
        pids.append(net.add_qp(Q, lin, [cid], [i], 0.0))
    net.add_edges([(pids[i], pids[j]) for (i, j) in edges])
    net.assign_constraint_groups()
    net.set_options(debug_visualize=False)
    net.default_initialization = np.zeros(n)
    return net


QUANTITY = [("Cournot, 2 firms", (2, 10.0, 1.0, 1.0), (), np.full(2, 3.0)),
            ("Cournot, 3 firms", (3, 10.0, 1.0, 1.0), (), np.full(3, 2.25)),
            ("Cournot, 5 firms, b = 2", (5, 13.0, 2.0, 1.0), (), np.full(5, 1.0)),
            ("Stackelberg, two levels", (2, 10.0, 1.0, 1.0), ((0, 1),), np.array([4.5, 2.25])),
            ("three-level chain", (3, 9.0, 1.0, 1.0), ((0, 1), (1, 2)), np.array([4.0, 2.0, 1.0])),
            ("a leader above two Nash followers", (3, 10.0, 1.0, 1.0), ((0, 1), (0, 2)), np.array([4.5, 1.5, 1.5]))]


def backward_substitution(net):
    """An independent statement of the equilibrium of a network of QUADRATIC players while no constraint is active: player j
    picks its own variables knowing the (affine) joint response of the sub-network of its descendants, everything else
    held fixed; the sub-network's response is the simultaneous solution of its members' first-order conditions, each with
    the total derivative through ITS descendants.  Plain linear algebra, no pivoting method, no polyhedra."""
    nv = len(net.default_initialization)
    pids = sorted(net.qps.keys())
    own = {i: sorted(net.qps[i].var_indices) for i in pids}

    def descendants(i, acc=None):
        acc = set() if acc is None else acc
        for c in net.network_edges[i]:
            if c not in acc:
                acc.add(c)
                descendants(c, acc)
        return acc

    memo = {}

    def response(D):
        """x[V(D)] = K x[rest] + k for the players in D."""
        if D in memo:
            return memo[D]
        VD = sorted(v for j in D for v in own[j])
        rest = [v for v in range(nv) if v not in VD]
        rows, rhs = [], []
        for j in sorted(D):
            Q = np.asarray(net.qps[j].f.Q, dtype=float)
            Q = 0.5 * (Q + Q.T)
            T = np.zeros((len(own[j]), nv))
            for a, v in enumerate(own[j]):
                T[a, v] = 1.0
            Dj = frozenset(descendants(j))
            if Dj:
                Kj, _, VDj, restj = response(Dj)
                for a, v in enumerate(own[j]):
                    T[a, VDj] = Kj[:, restj.index(v)]
            rows.append(T @ Q)
            rhs.append(T @ np.asarray(net.qps[j].f.q, dtype=float))
        Mx, c = np.vstack(rows), np.concatenate(rhs)
        K = -np.linalg.solve(Mx[:, VD], Mx[:, rest])
        k = -np.linalg.solve(Mx[:, VD], c)
        memo[D] = (K, k, VD, rest)
        return memo[D]

    _, k, VD, _ = response(frozenset(pids))
    x = np.zeros(nv)
    x[VD] = k
    return x


# every shape of hierarchy four players allow up to relabelling that the example accepts: flat, one to four levels,
# several players on a level, a child with two parents, two separate pairs
HIERARCHIES = [[], [(1, 2)], [(1, 2), (2, 3)], [(1, 2), (1, 3)], [(1, 3), (2, 3)], [(1, 2), (2, 3), (3, 4)],
               [(1, 2), (3, 4)], [(1, 2), (1, 3), (1, 4)], [(1, 2), (2, 3), (2, 4)], [(1, 3), (2, 3), (3, 4)]]


def _check_hierarchies(engine):
    from qpn_amd import examples
    for edges in HIERARCHIES:
        for seed in (1, 4):
            net = examples.setup("four_player_matrix_game", edge_list=edges, seed=seed)
            want = backward_substitution(net)
            assert np.max(np.abs(want)) < 5.0, "the box of the example must stay inactive for the closed form to hold"
            ret = algorithm.solve(net, engine=engine)
            assert ret["solved"], (edges, seed)
            assert np.max(np.abs(ret["x_opt"] - want)) <= 1e-9, (edges, seed, ret["x_opt"], want)


def _check(engine):
    for name, A, reg, want in GAMES:
        ret = algorithm.solve(matrix_game(A, reg), engine=engine)
        assert ret["solved"], name
        assert np.max(np.abs(ret["x_opt"] - want)) <= 1e-8, (name, ret["x_opt"])
    for name, par, edges, want in QUANTITY:
        ret = algorithm.solve(quantity_game(*par, edges=edges), engine=engine)
        assert ret["solved"], name
        assert np.max(np.abs(ret["x_opt"] - want)) <= 1e-8, (name, ret["x_opt"])


def test_analytic_equilibria_on_the_oracle_engine():
    from oracle_engine import OracleEngine
    _check(OracleEngine())


@pytest.mark.gpu
def test_analytic_equilibria_on_the_hip_engine(engine):
    _check(engine)


def test_four_player_hierarchies_against_backward_substitution_on_the_oracle_engine():
    from oracle_engine import OracleEngine
    _check_hierarchies(OracleEngine())


@pytest.mark.gpu
def test_four_player_hierarchies_against_backward_substitution_on_the_hip_engine(engine):
    _check_hierarchies(engine)
