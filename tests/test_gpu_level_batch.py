"""The level-wide batches of the outer loop on the HIP engine (level_batch.py): BASELINE config 4's
structure -- many independent leader-follower pairs in one two-level net -- through `algorithm.solve` itself, with O(1)
C-ABI calls per level and outer iteration, and the two new entry points behind it (qpn_recipes_batch, qpn_reduced_pieces)
against the CPU twin bit for bit.  CPU twin of the host logic: tests/test_level_batch.py."""
import collections

import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu

VERIFY = ("qpn_verify_nodes", "qpn_verify_nodes_h")
SOLVE = ("qpn_solve_nodes_into", "qpn_solve_nodes_h", "qpn_solve_avi_batch")


def _solved_nodes(engine, cnt, n, m, p, first=300):
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(first, cnt, n, m, p)
    w = P.shared_params(p)
    rec = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)
    res = engine.solve_nodes(*rec, w)
    assert np.all(np.asarray(res["status"]) == 1)
    return rec, np.asarray(res["active"]).astype(np.uint8)


@pytest.mark.parametrize("n,m,p", [(4, 5, 2), (16, 16, 8), (32, 32, 32), (7, 40, 1)])
def test_recipes_batch_and_reduced_pieces_equal_the_cpu_twin(engine, n, m, p):
    """Device == tests/oracle_engine.py bit for bit: the recipes of many solutions in one launch, and their local pieces with the
    multipliers eliminated (same operations in the same order, fp contraction off).  Host-pointer and device-pointer routes."""
    import torch
    from oracle_engine import OracleEngine
    cnt = 24
    rec, masks = _solved_nodes(engine, cnt, n, m, p)
    g = np.random.default_rng(n * 100 + m)
    for b in g.choice(cnt, 6, replace=False):                 # weakly active rows: two codes -> the node has several recipes
        for r in g.choice(m, min(2, m), replace=False):
            masks[b, n + r] |= (1 << 4) | (1 << 5)
    tot = np.array([int(np.prod([max(1, bin(int(v)).count("1")) for v in row])) for row in masks])
    cnts = np.minimum(tot, 3)
    cnts[1] = 0                                               # a node that asks for nothing
    off = np.concatenate([[0], np.cumsum(cnts)]).astype(np.int64)
    twin = OracleEngine()
    K0, no0 = twin.recipes_batch(masks, off)
    K1, no1 = engine.recipes_batch(masks, off)
    assert np.array_equal(K0, K1) and np.array_equal(no0, no1)
    dm = torch.tensor(masks, device="cuda:0")
    K2, no2 = engine.recipes_batch(dm, off)
    assert np.array_equal(K2.cpu().numpy(), K0) and np.array_equal(no2.cpu().numpy(), no0)
    want = twin.reduced_pieces(*rec, K0, no0)
    got = engine.reduced_pieces(*rec, K0, no0)
    names = ("Ar", "lr", "ur", "rows", "flags")
    for nm, a, b in zip(names, want, got):
        a = np.asarray(a); b = np.asarray(b)
        if nm in ("Ar", "lr", "ur"):                           # beyond rows[t] the twin leaves its fill values; compare the live part
            for t in range(len(want[3])):
                r = int(want[3][t])
                assert np.array_equal(a[t][..., :r], b[t][..., :r]), (nm, t)
        else:
            assert np.array_equal(a, b), nm
    dev = lambda a: torch.tensor(np.asarray(a), device="cuda:0")
    gd = engine.reduced_pieces(*[dev(a) for a in rec], dev(K0), dev(no0))
    torch.cuda.synchronize()
    for a, b in zip(got, gd):
        assert np.array_equal(np.asarray(a), b.cpu().numpy())


def test_recipes_batch_argument_errors(engine):
    from qpn_amd.engine import QpnError
    masks = np.full((3, 5), 2, np.uint8)
    with pytest.raises(QpnError):
        engine.recipes_batch(masks, np.array([0, 1, 3, 4], np.int64))          # node 1 has ONE recipe, asks for two
    with pytest.raises(QpnError):
        engine.recipes_batch(masks, np.array([1, 1, 1, 1], np.int64))          # offsets[0] != 0
    with pytest.raises(QpnError):
        engine.recipes_batch(masks, np.array([0, 1, 0, 1], np.int64))          # decreasing
    K, no = engine.recipes_batch(masks, np.array([0, 0, 0, 0], np.int64))
    assert K.shape == (0, 5)


def _run_pairs(engine, pairs, n, m):
    """solve() on the pairs net; returns (result, C-ABI call counts, number of process_level / solve_level sweeps)."""
    from qpn_amd import algorithm, examples, level_batch
    sweeps = collections.Counter()
    orig_p, orig_s = level_batch.process_level, level_batch.solve_level

    def proc(*a, **k):
        sweeps["process"] += 1
        return orig_p(*a, **k)

    def solv(*a, **k):
        sweeps["solve"] += 1
        return orig_s(*a, **k)

    algorithm.process_level = proc
    level_batch.solve_level = solv
    try:
        engine.calls.clear()
        net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
        ret = algorithm.solve(net, engine=engine)
    finally:
        algorithm.process_level = orig_p
        level_batch.solve_level = orig_s
    return ret, dict(engine.calls), sweeps


@pytest.mark.parametrize("pairs,n,m,sample", [(1000, 16, 16, 24), (200, 32, 32, 12)])
def test_pairs_net_through_solve(engine, pairs, n, m, sample):
    """VERDICT round 3, item 1's done-criterion: synthetic_pairs(pairs >= 1000, n = m = 16) and (200, n = m = 32) through
    algorithm.solve on the HIP engine: solved; x equal (1e-9) to every sampled pair solved ALONE on the CPU oracle engine;
    C-ABI calls per level sweep bounded by a constant that does not depend on the number of pairs."""
    from oracle_engine import OracleEngine
    from qpn_amd import algorithm, examples
    ret, calls, sweeps = _run_pairs(engine, pairs, n, m)
    assert ret["solved"], ret
    x = ret["x_opt"]
    g = np.random.default_rng(pairs)
    for k in sorted(g.choice(pairs, sample, replace=False).tolist()):
        one = algorithm.solve(examples.setup("synthetic_pairs", pairs=1, n=n, m=m, first=k), engine=OracleEngine())
        assert one["solved"], k
        assert np.max(np.abs(one["x_opt"] - x[2 * n * k:2 * n * (k + 1)])) <= 1e-9, k
    nver = sum(calls.get(k, 0) for k in VERIFY)
    nsol = sum(calls.get(k, 0) for k in SOLVE)
    # per process_level sweep: one verify per record shape (followers: 1; leaders: the piece rows are padded to multiples of 16,
    # a handful of shapes at most), one recipes call and one pieces call per shape; per solve_level sweep: one solve per shape --
    # plus the LP batches of remove_subsets / combine (node solves too): a constant per sweep, whatever the number of pairs
    assert nver <= 6 * sweeps["process"], (nver, sweeps)
    assert calls.get("qpn_recipes_batch", 0) <= 6 * sweeps["process"] and calls.get("qpn_reduced_pieces", 0) <= 6 * sweeps["process"]
    assert nsol <= 6 * (sweeps["process"] + sweeps["solve"]), (nsol, sweeps)
    small, calls_s, sweeps_s = _run_pairs(engine, 20, n, m)
    assert small["solved"]
    per = lambda c, s: sum(c.get(k, 0) for k in VERIFY + SOLVE + ("qpn_recipes_batch", "qpn_reduced_pieces")) / max(1, s["process"] + s["solve"])
    assert per(calls, sweeps) <= 2.0 * per(calls_s, sweeps_s) + 2.0                 # 10 - 50 x the nodes, the same calls per sweep


def test_pairs_net_oracle_and_hip_engines_agree(engine):
    from oracle_engine import OracleEngine
    from qpn_amd import algorithm, examples
    a = algorithm.solve(examples.setup("synthetic_pairs", pairs=16, n=6, m=9), engine=engine)
    b = algorithm.solve(examples.setup("synthetic_pairs", pairs=16, n=6, m=9), engine=OracleEngine())
    assert a["solved"] and b["solved"] and np.max(np.abs(a["x_opt"] - b["x_opt"])) <= 1e-9


def test_inert_rows_do_not_change_a_node_solve(engine, oracle):
    """Record batches pad missing constraint rows with 0'x in (-inf, inf): the solve and the verification of a padded record
    equal the unpadded one's (the padded multipliers are 0, masks code 6)."""
    from qpn_amd.engine import colmajor
    for n, m, mp in ((16, 9, 16), (32, 20, 32), (32, 33, 48), (40, 50, 64)):
        Q, R, qd, A, B, l, u = P.synth_nodes(40, 12, n, m, 4)
        w = P.shared_params(4)
        Ap = np.zeros((12, mp, n)); Ap[:, :m] = A
        Bp = np.zeros((12, mp, 4)); Bp[:, :m] = B
        lp = np.full((12, mp), -np.inf); lp[:, :m] = l
        up = np.full((12, mp), np.inf); up[:, :m] = u
        r0 = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
        r1 = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(Ap), colmajor(Bp), lp, up, w)
        assert np.all(np.asarray(r0["status"]) == 1) and np.all(np.asarray(r1["status"]) == 1)
        z0, z1 = np.asarray(r0["z"]), np.asarray(r1["z"])
        assert np.max(np.abs(z0[:, :n + m] - z1[:, :n + m])) <= 1e-9 and np.all(z1[:, n + m:] == 0.0)
        assert np.array_equal(np.asarray(r0["active"]), np.asarray(r1["active"])[:, :n + m])
        assert np.all(np.asarray(r1["active"])[:, n + m:] == (1 << 5))
        s0, lam0, p0 = engine.verify_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, z0[:, :n], w)
        s1, lam1, p1 = engine.verify_nodes(colmajor(Q), colmajor(R), qd, colmajor(Ap), colmajor(Bp), lp, up, z0[:, :n], w)
        assert np.array_equal(np.asarray(s0), np.asarray(s1)) and np.array_equal(np.asarray(p0), np.asarray(p1))
        assert np.max(np.abs(np.asarray(lam0) - np.asarray(lam1)[:, :m])) <= 1e-9
