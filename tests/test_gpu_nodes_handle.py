"""Resident node records (qpn_nodes_upload / qpn_solve_nodes_h / qpn_verify_nodes_h): the handle route gives what
qpn_solve_nodes_into / qpn_verify_nodes give with the records passed every call -- bit for bit, since the same kernels run
on the same data -- and the oracle's answer (status, active-set masks bit-exact, primals within 1e-9, the bar of
DESIGN.md section 2).  Also: what the handle learns from its records (no node needs the general kernel -> one launch per
sweep; some do -> they keep being re-solved), qpn_nodes_update, host- and device-resident callers."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def _records(seed, cnt, n, m, p=8):
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(seed, cnt, n, m, p)
    B = np.random.default_rng(seed).standard_normal((cnt, m, p)) * 0.1
    return (Q, R, qd, A, B, l, u), (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)


def _oracle(oracle, rec, w):
    M, q, lo, hi, kind = P.reduced_blocks(*rec, w)
    return oracle.solve_avi_batch(M, q, lo, hi, kind=kind)


class _general_variants:
    """Resident records whose Qd blocks are all symmetric take kernel variants of their own (QPN_OPT_SYM_ROUTE, n = m = 32):
    equal to the general ones up to rounding, not bit for bit.  The tests below that compare the handle route with the
    per-call route BIT FOR BIT (same kernels on the same data) switch the symmetric variants off for their duration."""
    def __init__(self, engine):
        self.engine = engine

    def __enter__(self):
        from qpn_amd import _lib
        self.engine.set_option(_lib.OPT_SYM_ROUTE, 0)

    def __exit__(self, *exc):
        from qpn_amd import _lib
        self.engine.set_option(_lib.OPT_SYM_ROUTE, 1)


def _same(res, ref, n, tol=1e-9):
    assert np.array_equal(np.asarray(res["status"]), ref["status"])
    assert np.array_equal(np.asarray(res["active"]), ref["active"])
    z, zr = np.asarray(res["z"]), ref["z"]
    assert np.max(np.abs(z - zr)) <= tol * max(1.0, np.max(np.abs(zr)))


@pytest.mark.parametrize("n,m", [(32, 32), (12, 20), (32, 7)])
def test_handle_host_route_equals_per_call_route_and_oracle(engine, oracle, n, m):
    rec, abi = _records(11, 96, n, m)
    nodes = engine.upload_nodes(*abi)
    rng = np.random.default_rng(1)
    for sweep in range(4):                       # new parameters every sweep, same records
        w = rng.standard_normal(8)
        with _general_variants(engine):
            a = nodes.solve(w)
        b = engine.solve_nodes(*abi, w)
        for k in ("z", "status", "resid", "pivots", "active"):
            assert np.array_equal(a[k], b[k]), k
        _same(a, _oracle(oracle, rec, w), n)
    info = nodes.info()
    assert info["decline_state"] == 2 and info["declined"] == 0      # strictly convex nodes: the fused kernel takes all
    nodes.close()


def test_handle_with_per_node_parameters_and_primal_blocks_only(engine, oracle):
    n, m, cnt = 32, 32, 64
    rec, abi = _records(5, cnt, n, m)
    nodes = engine.upload_nodes(*abi)
    W = np.random.default_rng(2).standard_normal((cnt, 8))
    x = np.zeros((cnt, 40))
    with _general_variants(engine):
        out = nodes.solve(W, want=(), x_out=x)                     # only statuses and the primal blocks come back
    assert out["z"] is None and np.all(out["status"] == 1)
    ref = engine.solve_nodes(*abi, W)
    assert np.array_equal(x[:, :n], ref["z"][:, :n]) and np.all(x[:, n:] == 0)
    M, q, lo, hi, kind = P.reduced_blocks(*rec, W)
    zr = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)["z"]
    assert np.max(np.abs(x[:, :n] - zr[:, :n])) <= 1e-9


def test_handle_device_route(engine, oracle):
    import torch
    n, m, cnt = 32, 32, 300
    rec, abi = _records(21, cnt, n, m)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    dabi = [t(a) for a in abi]
    nodes = engine.upload_nodes(*dabi)
    for a in dabi:                                    # the handle holds its own copy: the caller's buffers may go
        a.zero_()
    ring = t(np.random.default_rng(3).standard_normal((5, 8)))
    x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
    out = None
    for k in range(5):
        out = nodes.solve(ring[k], out=out, x_out=x)
        torch.cuda.synchronize()
        host = {kk: v.cpu().numpy() for kk, v in out.items()}
        _same(host, _oracle(oracle, rec, ring[k].cpu().numpy()), n)
        assert np.array_equal(x.cpu().numpy(), host["z"][:, :n])
        assert float(out["resid"].max()) <= 1e-8
    assert nodes.info()["decline_state"] == 2
    with pytest.raises(Exception):
        nodes.solve(ring[0].float())                  # wrong dtype is refused, not reinterpreted


def test_handle_keeps_resolving_nodes_that_need_the_general_kernel(engine, oracle):
    """Equality constraint rows and a node whose H block needs pivoting are declined by the fused kernel; the handle
    must learn that (state 3) and keep launching the general kernel for them, sweep after sweep."""
    n, m, cnt = 32, 32, 40
    rec, abi = _records(9, cnt, n, m)
    Q, R, qd, A, B, l, u = [a.copy() for a in rec]
    u[3, 5] = l[3, 5]                                  # equality row
    u[17, :4] = l[17, :4]
    Qs = Q[29].copy(); Qs[0, 0] = 1e-9; Qs[0, 1:] *= 1e-3; Qs[1:, 0] *= 1e-3      # tiny leading pivot
    Q[29] = 0.5 * (Qs + Qs.T) + np.diag([0.0] + [0.5] * (n - 1))
    from qpn_amd.engine import colmajor
    rec2 = (Q, R, qd, A, B, l, u)
    abi2 = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)
    nodes = engine.upload_nodes(*abi2)
    rng = np.random.default_rng(4)
    for sweep in range(4):
        w = rng.standard_normal(8)
        with _general_variants(engine):
            a = nodes.solve(w)
        b = engine.solve_nodes(*abi2, w)
        for k in ("z", "status", "pivots", "active"):
            assert np.array_equal(a[k], b[k]), k
        ref = _oracle(oracle, rec2, w)
        assert np.array_equal(a["status"], ref["status"]) and np.array_equal(a["active"], ref["active"])
        assert np.max(np.abs(a["z"] - ref["z"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["z"])))
    info = nodes.info()
    assert info["decline_state"] == 3 and info["declined"] >= 2
    # the same records through the symmetric variants (all Qd blocks are symmetric): the declined nodes take the general kernel
    # as before, everything matches the oracle
    assert info["symmetric"]
    w = rng.standard_normal(8)
    a = nodes.solve(w)
    ref = _oracle(oracle, rec2, w)
    assert np.array_equal(a["status"], ref["status"]) and np.array_equal(a["active"], ref["active"])
    assert np.max(np.abs(a["z"] - ref["z"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["z"])))
    assert nodes.info()["declined"] >= 2


def test_handle_update_replaces_a_field_and_forgets_what_it_knew(engine, oracle):
    n, m, cnt = 32, 32, 48
    rec, abi = _records(13, cnt, n, m)
    nodes = engine.upload_nodes(*abi)
    w = P.shared_params()
    nodes.solve(w); nodes.solve(w)
    assert nodes.info()["decline_state"] == 2
    Q, R, qd, A, B, l, u = rec
    u2 = u.copy(); u2[7, 2] = l[7, 2]                  # now one node has an equality row -> must be re-learned
    nodes.update("u", u2)
    assert nodes.info()["decline_state"] == 0
    a = nodes.solve(w)
    ref = _oracle(oracle, (Q, R, qd, A, B, l, u2), w)
    _same(a, ref, n)
    nodes.solve(w)
    assert nodes.info()["decline_state"] == 3 and nodes.info()["declined"] == 1


def test_handle_verify_equals_verify_nodes(engine, oracle):
    n, m, cnt = 20, 24, 50
    rec, abi = _records(31, cnt, n, m)
    nodes = engine.upload_nodes(*abi)
    w = P.shared_params()
    sol = nodes.solve(w)
    xd = sol["z"][:, :n].copy()
    xd[::3] += 0.05                                     # some nodes off their optimum
    s1, l1, p1 = nodes.verify(xd, w)
    s2, l2, p2 = engine.verify_nodes(*abi, xd, w)
    assert np.array_equal(s1, s2) and np.array_equal(p1, p2) and np.array_equal(l1, l2)
    Q, R, qd, A, B, l, u = rec
    for i in range(cnt):
        sc, lc, pc = oracle.verify_solution(Q[i], R[i], qd[i], A[i], B[i], l[i], u[i], xd[i], w)
        assert bool(s1[i]) == sc and p1[i] == pc


def test_handle_large_nodes_take_the_general_route(engine, oracle):
    """n, m beyond the fused kernel's 32: the handle still spares the record traffic; results as per call."""
    n, m, cnt = 40, 50, 6
    rec, abi = _records(41, cnt, n, m, p=3)
    nodes = engine.upload_nodes(*abi)
    w = P.shared_params(3)
    a = nodes.solve(w)
    b = engine.solve_nodes(*abi, w)
    for k in ("z", "status", "active"):
        assert np.array_equal(a[k], b[k]), k
    _same(a, _oracle(oracle, rec, w), n)


def test_handle_abi_errors(engine):
    import ctypes as C
    lib = engine.lib
    h = C.c_void_p()
    assert lib.qpn_nodes_upload(engine.ctx, 0, 4, 4, 1, None, None, None, None, None, None, None, 0, C.byref(h)) == -1
    st = (C.c_int32 * 4)()
    assert lib.qpn_solve_nodes_h(engine.ctx, None, None, 0, None, st, None, None, None, None, 0, None, 0) == -1
    assert lib.qpn_nodes_free(engine.ctx, None) == 0


def test_handle_fast_path_notices_swapped_or_dropped_output_buffers(engine, oracle):
    """The sweep loop's cached argument tail holds raw device addresses: a dict whose tensors were swapped or deleted, a w
    of another length, a mis-shaped caller buffer -- each must rebuild or raise, never write to a stale address."""
    import torch
    from qpn_amd.engine import QpnError
    n, m, p, cnt = 32, 32, 8, 300
    rec, abi = _records(11, cnt, n, m)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    dev = [t(a) for a in abi]
    w = t(P.shared_params(p))
    nodes = engine.upload_nodes(*dev)
    out = nodes.solve(w)
    out = nodes.solve(w, out=out)                   # second call: the fast path is armed with these buffers
    torch.cuda.synchronize()
    z_ref = out["z"].clone(); st_ref = out["status"].clone()
    # (1) a tensor of the dict is replaced: the next sweep must write into the NEW tensor and leave the old one alone
    old_z = out["z"]
    old_z.fill_(7.0)
    out["z"] = torch.zeros_like(old_z)
    nodes.solve(w, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out["z"], z_ref) and bool((old_z == 7.0).all())
    # (2) an optional output is dropped from the dict: nothing is written for it any more
    old_act = out.pop("active")
    old_act.fill_(9)
    nodes.solve(w, out=out)
    torch.cuda.synchronize()
    assert bool((old_act == 9).all()) and torch.equal(out["status"], st_ref)
    # (3) parameters of the wrong length or on the host while the buffers are on the device
    with pytest.raises(QpnError):
        nodes.solve(w[: p - 1], out=out)
    # (4) a caller-supplied buffer of the wrong shape / dtype is refused when first seen
    bad = dict(status=torch.zeros(cnt, dtype=torch.int32, device="cuda:0"), z=torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0"))
    with pytest.raises(QpnError):
        nodes.solve(w, out=bad)
    bad = dict(status=torch.zeros(cnt, dtype=torch.int64, device="cuda:0"))
    with pytest.raises(QpnError):
        nodes.solve(w, out=bad)
    nodes.close()


def _skewed(rec, eps=0.05):
    """The same records with a skew part added to every Qd: x'Qx is unchanged (still strictly convex), Q is not symmetric."""
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = rec
    K = np.random.default_rng(77).standard_normal(Q.shape) * eps
    Q2 = Q + (K - K.transpose(0, 2, 1))
    return (Q2, R, qd, A, B, l, u), (colmajor(Q2), colmajor(R), qd, colmajor(A), colmajor(B), l, u)


def test_symmetric_records_take_the_symmetric_variants(engine, oracle):
    """n = m = 32 with every Qd bitwise symmetric: the handle knows (info), the sweep runs avi_solve_schur<.., SYM> (the lower-left
    tile of H is never formed, S(1,0) is the transpose of S(0,1)).  Same statuses, masks and pivot counts as the general
    variant, primals equal to rounding; both within the bar against the oracle.  Also above one resident round (the
    staggered instantiation)."""
    from qpn_amd import _lib
    n, m = 32, 32
    for seed, cnt in ((31, 200), (32, 4500)):
        rec, abi = _records(seed, cnt, n, m)
        nodes = engine.upload_nodes(*abi)
        assert nodes.info()["symmetric"]
        rng = np.random.default_rng(seed)
        for sweep in range(3):
            w = rng.standard_normal(8)
            a = {k: np.array(v) for k, v in nodes.solve(w).items()}
            with _general_variants(engine):
                g = {k: np.array(v) for k, v in nodes.solve(w).items()}
            for k in ("status", "active", "pivots"):
                assert np.array_equal(a[k], g[k]), k
            assert np.max(np.abs(a["z"] - g["z"])) <= 1e-11 * max(1.0, np.max(np.abs(g["z"])))
            assert float(a["resid"].max()) <= 1e-8
            if cnt <= 200 or sweep == 0:
                sub = slice(0, 200)
                ref = _oracle(oracle, tuple(x[sub] for x in rec), w)
                _same({k: a[k][sub] for k in ("status", "active", "z")}, ref, n)
        nodes.close()
    assert engine.set_option(_lib.OPT_SYM_ROUTE, 1) is None


def test_asymmetric_records_take_the_general_variants(engine, oracle):
    """One skew part in the Qd blocks and the handle must not use the symmetric variants: bit for bit the per-call route's
    answer, and the oracle's within the bar (the AVI is still strictly monotone: the skew part does not change x'Qx)."""
    n, m, cnt = 32, 32, 150
    rec, _ = _records(41, cnt, n, m)
    rec2, abi2 = _skewed(rec)
    nodes = engine.upload_nodes(*abi2)
    assert not nodes.info()["symmetric"]
    w = np.random.default_rng(5).standard_normal(8)
    a = nodes.solve(w)
    b = engine.solve_nodes(*abi2, w)
    for k in ("z", "status", "resid", "pivots", "active"):
        assert np.array_equal(a[k], b[k]), k
    _same(a, _oracle(oracle, rec2, w), n)
    # a single asymmetric entry pair in ONE node out of many is enough
    rec3, abi3 = _records(42, cnt, n, m)
    Q3 = rec3[0].copy(); Q3[cnt - 1, 30, 31] += 1e-13
    from qpn_amd.engine import colmajor
    nodes3 = engine.upload_nodes(colmajor(Q3), *abi3[1:])
    assert not nodes3.info()["symmetric"]
    nodes.close(); nodes3.close()


def test_update_of_qd_settles_the_symmetry_again(engine, oracle):
    n, m, cnt = 32, 32, 64
    rec, abi = _records(43, cnt, n, m)
    rec2, abi2 = _skewed(rec)
    nodes = engine.upload_nodes(*abi)
    w = P.shared_params()
    assert nodes.info()["symmetric"]
    _same(nodes.solve(w), _oracle(oracle, rec, w), n)
    nodes.update("Qd", abi2[0])                        # (column-major, as uploaded)
    assert not nodes.info()["symmetric"]
    _same(nodes.solve(w), _oracle(oracle, rec2, w), n)
    nodes.update("Qd", abi[0])
    assert nodes.info()["symmetric"]
    _same(nodes.solve(w), _oracle(oracle, rec, w), n)
    nodes.close()
