"""Row F1 on the device: qpn_local_pieces (local_piece, src/avi_solutions.jl:400-496 with find_non_trivial :384-388, for the
per-node GAVI of process_solution_graph, src/avi.jl:447-477) and qpn_recipes_from_masks (all_Ks, :200-215).

* device == the oracle's restatement (oracle/qpn_oracle.c::qpo_local_piece), bit for bit: the piece is data movement
  (copies, negations, the recipe's bounds), so Ap, lp, up and keep must be identical;
* recipe enumeration == itertools.product over the rows' code sets;
* on strictly convex leaves, from the masks of a HIP solve: the device pieces describe the same sets as the host-only
  restatement tests/strict_pieces.py::local_pieces_strict (lambda eliminated by substitution there, kept as a coordinate here):
  membership agrees on points of the recipe's equality manifold, inside and outside the inequalities."""
import itertools

import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = np.inf


def _records(seed, cnt, n, m, p):
    Q, R, qd, A, B, l, u = P.synth_nodes(seed, cnt, n, m, max(p, 1))
    rng = np.random.default_rng(seed)
    if p == 0:
        R = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0))
    else:
        B = rng.standard_normal((cnt, m, p))
    # mixed bound kinds incl. one-sided, free and equal bounds
    kind = rng.integers(0, 5, size=l.shape)
    l = np.where(kind == 1, -INF, l); u = np.where(kind == 2, INF, u)
    l = np.where(kind == 3, -INF, l); u = np.where(kind == 3, INF, u)
    u = np.where(kind == 4, l, u)
    return Q, R, qd, A, B, l, u


@pytest.mark.parametrize("n,m,p", [(32, 32, 8), (5, 9, 3), (7, 0, 2), (3, 4, 0), (40, 33, 3), (120, 130, 2)])
def test_local_pieces_equal_the_oracle_restatement(engine, oracle, n, m, p):
    from qpn_amd.engine import colmajor
    cnt = 6 if n + m <= 64 else 2
    Q, R, qd, A, B, l, u = _records(100 + n, cnt, n, m, p)
    rng = np.random.default_rng(n * 31 + m)
    pieces = 3 * cnt
    node_of = rng.integers(0, cnt, size=pieces).astype(np.int32)
    K = np.concatenate([rng.integers(1, 5, size=(pieces, n)), rng.integers(5, 9, size=(pieces, m))], axis=1).astype(np.uint8)
    K[0, :] = 0                                            # code 0 (no condition) is accepted: treated as free
    Ap, lp, up, keep = engine.local_pieces(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, K, node_of=node_of)
    for t in range(pieces):
        b = node_of[t]
        Kt = K[t].copy()
        if t == 0:
            Kt[:n] = 4; Kt[n:] = 8
        Ao, lo, uo, ko = oracle.local_piece(Q[b], R[b], qd[b], A[b], B[b], l[b], u[b], Kt)
        assert np.array_equal(Ap[t].T, Ao) and np.array_equal(lp[t], lo) and np.array_equal(up[t], uo) and np.array_equal(keep[t], ko)
    # default node_of: piece t <-> node t
    Ap2, lp2, up2, keep2 = engine.local_pieces(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, K[:cnt])
    for t in range(cnt):
        Kt = K[t].copy()
        if t == 0:
            Kt[:n] = 4; Kt[n:] = 8
        Ao, lo, uo, ko = oracle.local_piece(Q[t], R[t], qd[t], A[t], B[t], l[t], u[t], Kt)
        assert np.array_equal(Ap2[t].T, Ao) and np.array_equal(lp2[t], lo) and np.array_equal(keep2[t], ko)


def test_local_pieces_device_buffers(engine, oracle):
    import torch
    from qpn_amd.engine import colmajor
    n, m, p, cnt = 12, 17, 4, 5
    Q, R, qd, A, B, l, u = _records(7, cnt, n, m, p)
    rng = np.random.default_rng(1)
    K = np.concatenate([rng.integers(1, 5, size=(cnt, n)), rng.integers(5, 9, size=(cnt, m))], axis=1).astype(np.uint8)
    t = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    Ap, lp, up, keep = engine.local_pieces(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u),
                                           t(K, torch.uint8))
    torch.cuda.synchronize()
    for b in range(cnt):
        Ao, lo, uo, ko = oracle.local_piece(Q[b], R[b], qd[b], A[b], B[b], l[b], u[b], K[b])
        assert np.array_equal(Ap[b].cpu().numpy().T, Ao) and np.array_equal(lp[b].cpu().numpy(), lo)
        assert np.array_equal(up[b].cpu().numpy(), uo) and np.array_equal(keep[b].cpu().numpy(), ko)


def test_local_pieces_bad_node_index(engine, oracle):
    """node_of outside 0 .. nodes-1: an argument error for host arrays; for DEVICE arrays (which the host does not read) the
    kernel itself returns an empty piece -- every row dropped, infinite bounds, zero coefficients -- instead of reading records
    that do not exist; the pieces next to it are untouched.  Codes outside a row's range mean "no condition" like code 0."""
    import torch
    from qpn_amd.engine import colmajor
    n, m, p, cnt = 6, 7, 2, 3
    Q, R, qd, A, B, l, u = _records(11, cnt, n, m, p)
    rng = np.random.default_rng(2)
    K = np.concatenate([rng.integers(1, 4, size=(4, n)), rng.integers(5, 9, size=(4, m))], axis=1).astype(np.uint8)
    node_of = np.array([0, 7, 2, -1], np.int32)
    args = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)
    with pytest.raises(Exception):
        engine.local_pieces(*args, K, node_of=node_of)
    t = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    Ap, lp, up, keep = engine.local_pieces(*[t(a) for a in args], t(K, torch.uint8), node_of=t(node_of, torch.int32))
    torch.cuda.synchronize()
    for tt in (1, 3):
        assert not keep[tt].any() and not Ap[tt].any()
        assert bool((lp[tt] == -INF).all()) and bool((up[tt] == INF).all())
    for tt in (0, 2):
        b = int(node_of[tt])
        Ao, lo, uo, ko = oracle.local_piece(Q[b], R[b], qd[b], A[b], B[b], l[b], u[b], K[tt])
        assert np.array_equal(Ap[tt].cpu().numpy().T, Ao) and np.array_equal(keep[tt].cpu().numpy(), ko)
        assert np.array_equal(lp[tt].cpu().numpy(), lo) and np.array_equal(up[tt].cpu().numpy(), uo)
    # code 0 and an out-of-range code: the same piece as the "free" codes (4 on x rows; nothing comparable on constraint rows:
    # there 0 leaves the row unconditioned)
    K0 = K[:1].copy(); K0[0, 0] = 0
    K4 = K[:1].copy(); K4[0, 0] = 4
    a0 = engine.local_pieces(*args, K0, node_of=np.zeros(1, np.int32))
    a4 = engine.local_pieces(*args, K4, node_of=np.zeros(1, np.int32))
    for x0, x4 in zip(a0, a4):
        assert np.array_equal(np.asarray(x0), np.asarray(x4))


def test_recipe_enumeration_is_the_cartesian_product(engine):
    rng = np.random.default_rng(3)
    for trial in range(5):
        N = int(rng.integers(1, 9))
        mask = np.zeros(N, np.uint8)
        for i in range(N):
            bits = rng.choice(4, size=int(rng.integers(1, 4)), replace=False) + (4 if i >= N // 2 else 0)
            mask[i] = sum(1 << int(b) for b in bits)
        K, total = engine.recipes_from_masks(mask)
        sets = [[c + 1 for c in range(8) if (int(mk) >> c) & 1] for mk in mask]
        want = {tuple(reversed(r)) for r in itertools.product(*reversed(sets))}       # row 0 fastest
        assert total == len(want) == K.shape[0]
        assert {tuple(int(v) for v in row) for row in K} == want
        first = [tuple(s[0] for s in sets)]
        assert tuple(int(v) for v in K[0]) == first[0]
        K2, _ = engine.recipes_from_masks(mask, first=total - 1, count=1)
        assert tuple(int(v) for v in K2[0]) == tuple(s[-1] for s in sets)
    with pytest.raises(Exception):
        engine.recipes_from_masks(mask, first=total, count=1)


def _code(opt):
    return {"lo": 5, "in": 6, "up": 7, "eq": 8}[opt]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_device_pieces_describe_the_sets_of_the_strict_restatement(engine, seed):
    """A strictly convex leaf (Q_dd > 0, independent active rows): solve on the HIP path, take the masks, and for every
    recipe compare the device piece over [x_d; lambda; x_p] with local_pieces_strict's piece over x."""
    from qpn_amd import avi_solutions as AS
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(seed)
    nv, n, m = 7, 4, 5
    dec = sorted(int(v) for v in rng.choice(nv, size=n, replace=False))
    par = [i for i in range(nv) if i not in dec]
    G = rng.standard_normal((nv, nv)); Q = G @ G.T / nv + 0.5 * np.eye(nv)
    q = rng.standard_normal(nv)
    A = rng.standard_normal((m, nv)) / 2
    l = -0.3 - 0.5 * rng.random(m); u = 0.3 + 0.5 * rng.random(m)
    x0 = rng.standard_normal(nv) * 0.5
    rec = AS.node_records(Q, q, A, l, u, dec)
    w = x0[par]
    res = engine.solve_nodes(colmajor(rec["Qd"][None]), colmajor(rec["R"][None]), rec["qd"][None], colmajor(rec["Ad"][None]),
                             colmajor(rec["B"][None]), l[None], u[None], w)
    assert res["status"][0] == 1
    z = res["z"][0]; mask = res["active"][0]
    x = x0.copy(); x[dec] = z[:n]; lam = z[n:]
    K_all, total = AS.all_Ks(mask, engine=engine)
    assert total == K_all.shape[0] >= 1 and np.all(K_all[:, :n] == 2)              # free x_d rows: code 2 only
    from strict_pieces import local_pieces_strict
    strict = local_pieces_strict(Q, q, A, l, u, dec, x, lam)
    assert len(strict) >= 1
    dev = AS.local_pieces(rec, K_all, engine=engine)
    zw = np.concatenate([z, w])
    assert any(Pd.contains(zw, tol=1e-6) for Pd in dev)                             # the solution lies in one of its pieces
    checked = 0
    for rec_opts in itertools.product(*[[o for o in ("lo", "up", "in")] for _ in range(m)]):
        K = np.array([2] * n + [_code(o) for o in rec_opts], np.uint8)
        if not any(np.array_equal(K, row) for row in K_all):
            continue                                                               # not compatible with the masks
        Pd = AS.local_pieces(rec, K[None], engine=engine)[0]
        act = [i for i in range(m) if rec_opts[i] != "in"]
        bnd = np.array([l[i] if rec_opts[i] == "lo" else u[i] for i in act])
        # the strict piece of the same recipe (built by the same routine from a point that satisfies it exactly)
        Aa = rec["Ad"][act]; na = len(act)
        KKT = np.block([[rec["Qd"], -Aa.T], [Aa, np.zeros((na, na))]])
        for scale in (0.0, 1e-3, 0.05, 0.5):
            wp = w + scale * rng.standard_normal(len(par))
            rhs = np.concatenate([-rec["qd"] - rec["R"] @ wp, bnd - rec["B"][act] @ wp])
            try:
                sol = np.linalg.solve(KKT, rhs)
            except np.linalg.LinAlgError:
                break
            xd = sol[:n]; lam_p = np.zeros(m); lam_p[act] = sol[n:]
            xp = np.zeros(nv); xp[dec] = xd; xp[par] = wp
            in_dev = Pd.contains(np.concatenate([xd, lam_p, wp]), tol=1e-7)
            # the set the recipe describes, stated directly: stationarity holds by construction; signs and inactive rows
            ax = A @ xp
            ok = True
            for i in range(m):
                if rec_opts[i] == "lo":
                    ok &= lam_p[i] >= -1e-7
                elif rec_opts[i] == "up":
                    ok &= lam_p[i] <= 1e-7
                else:
                    ok &= (l[i] - 1e-7 <= ax[i] <= u[i] + 1e-7)
            assert in_dev == bool(ok), (rec_opts, scale)
            checked += 1
            # and the host-only restatement agrees where it produced this recipe's piece
            for Ps in strict:
                if Ps.contains(xp, tol=1e-7) and scale == 0.0:
                    assert any(Pq.contains(np.concatenate([xd, lam_p, wp]), tol=1e-6) for Pq in dev)
    assert checked >= 4


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_solution_graph_pieces_equal_the_strict_restatement_as_sets(engine, seed):
    """The live loop's piece generator (masks -> qpn_recipes_from_masks -> qpn_local_pieces -> multipliers eliminated through
    the piece's own equalities) against the host-only restatement on strictly convex leaves: the same NUMBER of pieces, and
    the same SETS -- membership agrees on the solution, on points moved inside each piece and on points pushed across its
    faces."""
    from qpn_amd import avi_solutions as AS
    from qpn_amd.engine import colmajor
    from strict_pieces import local_pieces_strict
    rng = np.random.default_rng(100 + seed)
    nv, n, m = 8, 5, 6
    dec = sorted(int(v) for v in rng.choice(nv, size=n, replace=False))
    par = [i for i in range(nv) if i not in dec]
    G = rng.standard_normal((nv, nv)); Q = G @ G.T / nv + 0.5 * np.eye(nv)
    q = rng.standard_normal(nv)
    A = rng.standard_normal((m, nv)) / 2
    x0 = rng.standard_normal(nv) * 0.5
    l = A @ x0 - 0.05 - 0.3 * rng.random(m); u = A @ x0 + 0.05 + 0.3 * rng.random(m)      # x0 is feasible: the node has a solution
    rec = AS.node_records(Q, q, A, l, u, dec)
    res = engine.solve_nodes(colmajor(rec["Qd"][None]), colmajor(rec["R"][None]), rec["qd"][None], colmajor(rec["Ad"][None]),
                             colmajor(rec["B"][None]), l[None], u[None], x0[par])
    assert res["status"][0] == 1
    x = x0.copy(); x[dec] = res["z"][0][:n]; lam = res["z"][0][n:]
    dev = AS.solution_graph_pieces(Q, q, A, l, u, dec, x, lam, engine=engine)
    strict = local_pieces_strict(Q, q, A, l, u, dec, x, lam)
    assert len(dev) == len(strict) >= 1
    assert all(P.contains(x, tol=1e-6) for P in dev)
    pts = [x] + [x + s * rng.standard_normal(nv) for s in (1e-3, 1e-2, 0.1, 0.5) for _ in range(6)]
    for y in pts:
        in_dev = any(P.contains(y, tol=1e-7) for P in dev)
        in_str = any(P.contains(y, tol=1e-7) for P in strict)
        near = any(P.contains(y, tol=1e-5) for P in dev) != any(P.contains(y, tol=1e-9) for P in dev)   # on a face: skip
        assert in_dev == in_str or near, y
