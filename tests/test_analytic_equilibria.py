"""Analytic known answers through the whole host loop (solve -> solve_base -> process_qp / verify_solution -> solve_qep ->
pool assembly -> AVI solve): two-player zero-sum matrix games as single-level QPNets -- each player a QP over its mixed strategy
(simplex: one equality row + non-negativity), bilinear cost x'Ay / -x'Ay, i.e. degenerate (Q_dd = 0) nodes whose pool AVI is
monotone but not strictly so.  The mixed equilibria are textbook: rock-paper-scissors (1/3, 1/3, 1/3), matching pennies (1/2, 1/2),
the 2 x 2 game [[a, b], [c, d]] without a saddle point p = (d - c) / (a - b - c + d), q = (d - b) / (a - b - c + d).
And quantity competition with linear demand (price a - b sum q, unit cost c; player i: min  b q_i^2 + b q_i sum_{j != i} q_j - (a - c) q_i,
q_i >= 0) in its classical variants, on one, two and three levels of the network: Cournot-Nash with n firms q_i = (a - c) / ((n + 1) b);
Stackelberg leader-follower (a - c) / (2b), (a - c) / (4b); a three-level chain (a - c) / (2b), / (4b), / (8b); a leader above
two Nash followers (a - c) / (2b), then (a - c) / (6b) each.
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
