"""ctypes binding of the CPU oracle (oracle/qpn_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never from the product package.  The oracle is a CPU
restatement of the reference's node-AVI path ("parity unpinned" at the PATHSolver.solve_mcp
boundary, see oracle/qpn_oracle.h); it is the checker, never the thing shipped.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqpn_oracle.so")

SUCCESS, RAY_TERM, MAX_ITERS, FAILURE = 1, 2, 3, 4
ROW_STD, ROW_GAVI = 0, 1


class Opts(C.Structure):
    _fields_ = [("check_tol", C.c_double), ("piv_tol", C.c_double),
                ("feas_tol", C.c_double), ("max_pivots", C.c_int)]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only, no reference sources)."""
    src = os.path.join(_HERE, "qpn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.qpo_solve_avi.restype = C.c_int
        _lib.qpo_solve_avi_batch.restype = C.c_int
        _lib.qpo_check_avi_solution.restype = C.c_int
        _lib.qpo_natural_residual.restype = C.c_double
        _lib.qpo_verify_solution.restype = C.c_int
        _lib.qpo_num_threads.restype = C.c_int
    return _lib


def _p(a, t=C.c_double):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _colmajor(Mx):
    """numpy (rows, cols) matrix -> flat column-major buffer (Julia layout)."""
    return np.ascontiguousarray(np.asarray(Mx, dtype=np.float64).T).ravel()


def default_opts() -> Opts:
    o = Opts()
    lib().qpo_default_opts(C.byref(o))
    return o


def num_threads() -> int:
    """Threads the oracle should use: OpenMP's count capped by the affinity mask and the cgroup CPU quota (more
    threads than the quota only burst and are then throttled)."""
    import os
    c = int(lib().qpo_num_threads())
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            c = min(c, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, c)


def check_avi_solution(M, q, l, u, z, kind=None, tol=1e-6):
    """src/avi.jl:148-156 -> (sol_bad, degree, r)."""
    M = np.asarray(M, dtype=np.float64)
    N = M.shape[0]
    Mc, q, l, u, z = _colmajor(M), _f64(q), _f64(l), _f64(u), _f64(z)
    k = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
    r = np.empty(N)
    deg = lib().qpo_check_avi_solution(N, _p(Mc), _p(q), _p(l), _p(u), _p(k, C.c_uint8), _p(z),
                                       C.c_double(tol), _p(r))
    return deg > 0, int(deg), r


def natural_residual(M, q, l, u, z, kind=None):
    M = np.asarray(M, dtype=np.float64)
    N = M.shape[0]
    Mc, q, l, u, z = _colmajor(M), _f64(q), _f64(l), _f64(u), _f64(z)
    k = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
    return float(lib().qpo_natural_residual(N, _p(Mc), _p(q), _p(l), _p(u), _p(k, C.c_uint8), _p(z)))


def solve_avi(M, q, l, u, z0=None, kind=None, opts=None):
    """One AVI.  Returns dict(z, status, resid, pivots, active)."""
    M = np.asarray(M, dtype=np.float64)
    N = M.shape[0]
    Mc, q, l, u = _colmajor(M), _f64(q), _f64(l), _f64(u)
    z = np.zeros(N) if z0 is None else _f64(z0).copy()
    k = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
    res = C.c_double(0)
    piv = C.c_int(0)
    act = np.zeros(N, dtype=np.uint8)
    o = opts if opts is not None else default_opts()
    st = lib().qpo_solve_avi(N, _p(Mc), _p(q), _p(l), _p(u), _p(k, C.c_uint8), _p(z), C.byref(o),
                             C.byref(res), C.byref(piv), _p(act, C.c_uint8))
    return dict(z=z, status=int(st), resid=res.value, pivots=piv.value, active=act)


def solve_avi_batch(M, q, l, u, z0=None, kind=None, opts=None, nthreads=0):
    """Batch.  M: (batch, N, N) numpy row/col indexing, or (N, N) shared.  kind: (N,) or (batch, N)."""
    q = _f64(q)
    batch, N = q.shape
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 2:
        Mc = _colmajor(M)
        strideM = 0
    else:
        Mc = np.ascontiguousarray(np.transpose(M, (0, 2, 1))).ravel()
        strideM = N * N
    l, u = _f64(l), _f64(u)
    z = np.zeros((batch, N)) if z0 is None else _f64(z0).copy()
    if kind is None:
        k, sk = None, 0
    else:
        k = np.ascontiguousarray(kind, dtype=np.uint8)
        sk = 0 if k.ndim == 1 else N
    status = np.zeros(batch, dtype=np.int32)
    resid = np.zeros(batch)
    pivots = np.zeros(batch, dtype=np.int32)
    active = np.zeros((batch, N), dtype=np.uint8)
    o = opts if opts is not None else default_opts()
    nfail = lib().qpo_solve_avi_batch(batch, N, _p(Mc), C.c_long(strideM), _p(q), _p(l), _p(u),
                                      _p(k, C.c_uint8), C.c_long(sk), _p(z), C.byref(o),
                                      _p(status, C.c_int32), _p(resid), _p(pivots, C.c_int32),
                                      _p(active, C.c_uint8), int(nthreads) or num_threads())
    return dict(z=z, status=status, resid=resid, pivots=pivots, active=active, nfail=int(nfail))


def solve_avi_batch_colmajor(Mc, strideM, q, l, u, z0, kind, stride_kind, nthreads=0, opts=None):
    """Same, on buffers already in the C-ABI layout (column-major M); used by bench.py."""
    batch, N = q.shape
    z = _f64(z0).copy()
    status = np.zeros(batch, dtype=np.int32)
    resid = np.zeros(batch)
    pivots = np.zeros(batch, dtype=np.int32)
    o = opts if opts is not None else default_opts()
    nfail = lib().qpo_solve_avi_batch(batch, N, _p(Mc), C.c_long(strideM), _p(q), _p(l), _p(u),
                                      _p(kind, C.c_uint8), C.c_long(stride_kind), _p(z),
                                      C.byref(o), _p(status, C.c_int32), _p(resid),
                                      _p(pivots, C.c_int32), None, int(nthreads) or num_threads())
    return dict(z=z, status=status, resid=resid, pivots=pivots, nfail=int(nfail))


def convert_gavi(M, o, l1, u1, A, bw, l2, u2):
    """src/avi.jl:113-128, dense.  M: (d1, d1+d2), A: (d2, d1+d2); bw = B*w."""
    M = np.asarray(M, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64).reshape(-1, M.shape[1])
    d1, d2 = M.shape[0], A.shape[0]
    N = d1 + 2 * d2
    Mo = np.empty(N * N)
    qo, lo, uo = np.empty(N), np.empty(N), np.empty(N)
    lib().qpo_convert_gavi(d1, d2, _p(_colmajor(M)), _p(_f64(o)), _p(_f64(l1)), _p(_f64(u1)),
                           _p(_colmajor(A)), _p(_f64(bw)), _p(_f64(l2)), _p(_f64(u2)),
                           _p(Mo), _p(qo), _p(lo), _p(uo))
    return Mo.reshape(N, N).T.copy(), qo, lo, uo


def assemble_node(Qd, R, qd, Ad, B, l, u, w):
    """Reduced single-node KKT blocks.  Returns M, q, l, u, kind."""
    Qd = np.asarray(Qd, dtype=np.float64)
    n = Qd.shape[0]
    Ad = np.asarray(Ad, dtype=np.float64).reshape(-1, n)
    m = Ad.shape[0]
    w = _f64(w)
    p = w.shape[0]
    R = np.asarray(R, dtype=np.float64).reshape(n, p)
    B = np.asarray(B, dtype=np.float64).reshape(m, p)
    N = n + m
    Mo = np.empty(N * N)
    qo, lo, uo = np.empty(N), np.empty(N), np.empty(N)
    kind = np.empty(N, dtype=np.uint8)
    lib().qpo_assemble_node(n, m, p, _p(_colmajor(Qd)), _p(_colmajor(R)), _p(_f64(qd)),
                            _p(_colmajor(Ad)), _p(_colmajor(B)), _p(_f64(l)), _p(_f64(u)), _p(w),
                            _p(Mo), _p(qo), _p(lo), _p(uo), _p(kind, C.c_uint8))
    return Mo.reshape(N, N).T.copy(), qo, lo, uo, kind


def comp_indices(zv, rv, l, u, tol=1e-2, shift=0):
    """src/avi_solutions.jl:511-562 -> uint8 mask per row."""
    zv, rv, l, u = _f64(zv), _f64(rv), _f64(l), _f64(u)
    n = zv.shape[0]
    mask = np.zeros(n, dtype=np.uint8)
    lib().qpo_comp_indices(n, _p(zv), _p(rv), _p(l), _p(u), C.c_double(tol), int(shift),
                           _p(mask, C.c_uint8))
    return mask


def verify_solution(Qd, R, qd, Ad, B, l, u, xd, w, tol=1e-4):
    """src/qp_processing.jl:57-149 on a dense node record -> (solution, lambda, path)."""
    Qd = np.asarray(Qd, dtype=np.float64)
    n = Qd.shape[0]
    Ad = np.asarray(Ad, dtype=np.float64).reshape(-1, n)
    m = Ad.shape[0]
    w = _f64(w)
    p = w.shape[0]
    R = np.asarray(R, dtype=np.float64).reshape(n, p)
    B = np.asarray(B, dtype=np.float64).reshape(m, p)
    lam = np.zeros(max(m, 1))
    path = C.c_int(0)
    sol = lib().qpo_verify_solution(n, m, p, _p(_colmajor(Qd)), _p(_colmajor(R)), _p(_f64(qd)),
                                    _p(_colmajor(Ad)), _p(_colmajor(B)), _p(_f64(l)), _p(_f64(u)),
                                    _p(_f64(xd)), _p(w), C.c_double(tol), _p(lam), C.byref(path))
    return bool(sol), lam[:m].copy(), int(path.value)


def local_piece(Qd, R, qd, Ad, B, l, u, K):
    """qpo_local_piece: the piece of recipe K (codes 1..8 per row of z = [x_d; lambda]) before simplify.
    Math-layout inputs; returns (Ap (2N, N+p), lp, up, keep)."""
    Qd = _f64(Qd); n = Qd.shape[0]
    Ad = _f64(Ad).reshape(-1, n); m = Ad.shape[0]
    R = _f64(R).reshape(n, -1); p = R.shape[1]
    B = _f64(B).reshape(m, p)
    N = n + m
    K = np.ascontiguousarray(K, dtype=np.uint8)
    Ap = np.zeros(2 * N * (N + p)); lp = np.zeros(2 * N); up = np.zeros(2 * N); keep = np.zeros(2 * N, dtype=np.uint8)
    lib().qpo_local_piece(n, m, p, _p(_colmajor(Qd)), _p(_colmajor(R)), _p(_f64(qd)), _p(_colmajor(Ad)), _p(_colmajor(B)),
                          _p(_f64(l)), _p(_f64(u)), _p(K, C.c_uint8), _p(Ap), _p(lp), _p(up), _p(keep, C.c_uint8))
    return Ap.reshape(N + p, 2 * N).T.copy(), lp, up, keep
