"""Stage B of the large-node route by block principal pivoting (csrc/qpn_avi_schur_big.hip: schur_big_bpp; resident records with
bitwise symmetric Qd blocks, QPN_OPT_SYM_ROUTE = 1) on small batches of the class (64 < n <= 256, m <= 256):
* against the CPU oracle: status, active-set masks, primals and multipliers (1e-9 relative) -- nodes with one-sided rows, with tight
  two-sided boxes (most rows at a bound: the active set is capped at 112 rows) and nearly unconstrained ones;
* with the option off the handle runs the per-call route's kernels: bit for bit its answer (the Lemke kernel alone);
* a caller-set pivot budget keeps the Lemke kernel, whose pivots that budget counts: the oracle's pivot counts come back;
* an equality row makes the kernel hand the node to the Lemke kernel's route (decline / general path): solved all the same.
The full-size batch (512 x 256 x 256) is tests/test_gpu_config5_fullsize.py."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def _records(seed, cnt, n, m, mode):
    g = np.random.default_rng(seed)
    Q, R, qd, A, B, l, u = P.synth_nodes(30_000 + seed, cnt, n, m)
    w = P.shared_params()
    if mode == "tight":
        x0 = g.standard_normal((cnt, n))
        s0 = np.einsum("bij,bj->bi", A, x0) + B @ w
        l = s0 - g.uniform(0.0, 0.3, s0.shape); u = s0 + g.uniform(0.0, 0.3, s0.shape)
    elif mode == "loose":
        l = l - 3.0; u = u + 3.0
    return (Q, R, qd, A, B, l, u), w


@pytest.mark.parametrize("n,m,mode", [(96, 200, "plain"), (200, 150, "tight"), (256, 256, "tight"), (130, 40, "loose"), (70, 255, "plain")])
def test_block_principal_pivoting_equals_the_oracle(engine, oracle, n, m, mode):
    from qpn_amd import _lib
    from qpn_amd.engine import colmajor
    cnt = 5
    (Q, R, qd, A, B, l, u), w = _records(n + m, cnt, n, m, mode)
    M, q, lo, hi, kd = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kd)
    assert np.all(rc["status"] == 1)
    abi = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)
    per_call = {k: np.array(v) for k, v in engine.solve_nodes(*abi, w).items()}
    h = engine.upload_nodes(*abi)
    assert h.info()["symmetric"]
    h.solve(w)
    sym = {k: np.array(v) for k, v in h.solve(w).items()}
    assert np.array_equal(sym["status"], rc["status"]) and np.array_equal(sym["active"], rc["active"])
    scale = max(1.0, np.max(np.abs(rc["z"])))
    assert np.max(np.abs(sym["z"] - rc["z"])) <= 1e-9 * scale and np.max(sym["resid"]) <= 1e-8
    assert np.all(sym["pivots"] >= n)                       # n crash pivots + the complementarity pairs switched
    # the option off: the per-call route's kernels, bit for bit
    engine.set_option(_lib.OPT_SYM_ROUTE, 0)
    try:
        off = {k: np.array(v) for k, v in h.solve(w).items()}
    finally:
        engine.set_option(_lib.OPT_SYM_ROUTE, 1)
    for k in ("z", "status", "active", "pivots"):
        assert np.array_equal(off[k], per_call[k]), k
    assert np.array_equal(per_call["pivots"], rc["pivots"])
    # a pivot budget belongs to the Lemke kernel
    og = engine.default_opts(); og.max_pivots = 100_000
    oo = oracle.default_opts(); oo.max_pivots = 100_000
    bud = {k: np.array(v) for k, v in h.solve(w, opts=og).items()}
    assert np.array_equal(bud["pivots"], oracle.solve_avi_batch(M, q, lo, hi, kind=kd, opts=oo)["pivots"])
    og.max_pivots = n + 3; oo.max_pivots = n + 3                # too small for most: MAX_ITERS exactly where the oracle says so
    few = {k: np.array(v) for k, v in h.solve(w, opts=og).items()}
    assert np.array_equal(few["status"], oracle.solve_avi_batch(M, q, lo, hi, kind=kd, opts=oo)["status"])
    h.close()


def test_equality_rows_are_left_to_the_other_route(engine, oracle):
    from qpn_amd.engine import colmajor
    n, m, cnt = 100, 140, 4
    (Q, R, qd, A, B, l, u), w = _records(9, cnt, n, m, "plain")
    g = np.random.default_rng(1)
    x0 = g.standard_normal((cnt, n))
    s0 = np.einsum("bij,bj->bi", A, x0) + B @ w
    l[:, :5] = u[:, :5] = s0[:, :5]                             # five equality rows through a common point
    l[:, 5:] = np.minimum(l[:, 5:], s0[:, 5:] - 0.1); u[:, 5:] = np.maximum(u[:, 5:], s0[:, 5:] + 0.1)
    M, q, lo, hi, kd = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kd)
    h = engine.upload_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)
    h.solve(w)
    res = {k: np.array(v) for k, v in h.solve(w).items()}
    h.close()
    assert np.array_equal(res["status"], rc["status"]) and np.all(res["status"] == 1)
    assert np.max(np.abs(res["z"] - rc["z"])) <= 1e-9 * max(1.0, np.max(np.abs(rc["z"])))
