"""Host wrapper over the C-ABI (include/qpn_hip.h): one ``Engine`` = one ``qpn_ctx`` on one GPU.

Buffers may be numpy arrays (host; the library stages them through HBM) or torch CUDA tensors
(device; zero-copy, asynchronous on torch's current stream).  All matrix buffers are in the
ABI layout: per item COLUMN-MAJOR (Julia), i.e. a ``(batch, N, N)`` array ``Mc`` holds
``Mc[b, j, i] = M_b[i, j]``.  ``colmajor()`` converts from the usual math layout.

There is no CPU path here: every method ends in a HIP kernel launch or raises.
"""
from __future__ import annotations

import collections
import ctypes as C
import time

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, AviOpts

try:  # torch is plumbing (device memory, streams, torch.distributed), not a requirement to import
    import torch
except Exception:  # pragma: no cover
    torch = None


class QpnError(RuntimeError):
    pass


def colmajor(M):
    """(..., rows, cols) math-layout array -> same data in the ABI's column-major layout."""
    if torch is not None and isinstance(M, torch.Tensor):
        return M.transpose(-1, -2).contiguous()
    return np.ascontiguousarray(np.swapaxes(np.asarray(M, dtype=np.float64), -1, -2))


def _torch_dt(dt):
    return {np.dtype(np.int32): torch.int32, np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8}[np.dtype(dt)]


def _is_dev(x):
    return torch is not None and isinstance(x, torch.Tensor) and x.is_cuda


def _ptr(x):
    if x is None:
        return None
    if torch is not None and isinstance(x, torch.Tensor):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


class Engine:
    """One context on one GPU.  ``device`` is a HIP device ordinal."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        self.lib = _lib.load_library()
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.qpn_ctx_create(self.device, C.byref(h))
        if rc != 0:
            raise QpnError(f"qpn_ctx_create({device}) failed: {self.lib.qpn_strerror(rc).decode()}")
        self.ctx = h
        self.use_torch_stream = use_torch_stream
        self.calls = collections.Counter()       # C-ABI calls per entry point (the tests assert O(1) calls per level with it)
        self.seconds = collections.Counter()     # wall time per entry point, method entry to the library's return (host-pointer
        self._t0 = None                          # calls are synchronous, so this is staging + kernels + read-back)
        self._bound = "own"                      # which stream the context launches on (a new context: its own)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.qpn_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _chk(self, rc, what):
        self.calls[what] += 1
        if self._t0 is not None:
            self.seconds[what] += time.perf_counter() - self._t0
            self._t0 = None
        if rc != 0:
            msg = self.lib.qpn_ctx_last_error(self.ctx).decode()
            raise QpnError(f"{what}: {self.lib.qpn_strerror(rc).decode()} ({msg})")

    def _bind_stream(self, dev):
        # device buffers: launch on torch's current stream so torch ops before/after the call are
        # ordered with the kernels (cuda_stream == 0 is the legacy default stream, a real stream)
        if dev and self.use_torch_stream:
            s = torch.cuda.current_stream(self.device).cuda_stream
            if s != self._bound:                 # (the binding call is skipped while the stream stays what it was)
                # the binding is cached only once the library has accepted it (the call synchronises the old stream and can fail)
                self._bound = None
                self._chk(self.lib.qpn_ctx_set_stream(self.ctx, C.c_void_p(s)), "qpn_ctx_set_stream")
                self._bound = s
        elif not dev:
            if self._bound != "own":
                self._bound = None
                self._chk(self.lib.qpn_ctx_use_own_stream(self.ctx), "qpn_ctx_use_own_stream")
                self._bound = "own"
        self._t0 = time.perf_counter()

    def synchronize(self):
        self._chk(self.lib.qpn_ctx_synchronize(self.ctx), "qpn_ctx_synchronize")

    def default_opts(self) -> AviOpts:
        o = AviOpts()
        self.lib.qpn_avi_default_opts(C.byref(o))
        return o

    def _mode(self, *arrs):
        devs = [_is_dev(a) for a in arrs if a is not None]
        if any(devs) and not all(devs):
            raise QpnError("mixing host and device buffers in one call")
        return bool(devs and devs[0])

    def _host(self, a, dtype):
        return None if a is None else np.ascontiguousarray(a, dtype=dtype)

    @staticmethod
    def _require_dev64(*ts):
        """Device arguments go to the kernels as raw addresses: they must be what the ABI says they are."""
        for t in ts:
            if t is None:
                continue
            if t.dtype != torch.float64 or not t.is_contiguous():
                raise QpnError("device buffers must be contiguous float64 tensors")

    @staticmethod
    def _x_stride(x_out, dev, batch, n):
        if x_out is None:
            return 0
        if dev:
            if x_out.dtype != torch.float64 or x_out.dim() != 2 or x_out.shape[0] != batch or x_out.shape[1] < n \
                    or x_out.stride(1) != 1:
                raise ValueError("x_out must be a [batch, >= n] fp64 tensor with unit inner stride")
            return x_out.stride(0)
        if x_out.dtype != np.float64 or x_out.ndim != 2 or x_out.shape[0] != batch or x_out.shape[1] < n \
                or x_out.strides[1] != 8:
            raise ValueError("x_out must be a [batch, >= n] float64 array with unit inner stride")
        return x_out.strides[0] // 8

    def upload_nodes(self, Qc, Rc, qd, Ac, Bc, l, u) -> "Nodes":
        """Make a level's node records resident (qpn_nodes_upload); see ``Nodes``."""
        return Nodes(self, Qc, Rc, qd, Ac, Bc, l, u)

    def _alloc(self, dev, shape, dtype):
        if dev:
            tdt = {np.float64: torch.float64, np.int32: torch.int32, np.uint8: torch.uint8}[dtype]
            return torch.empty(shape, dtype=tdt, device=f"cuda:{self.device}")
        return np.empty(shape, dtype=dtype)

    # -- (A2+A3+A9) ------------------------------------------------------------------------
    def solve_avi_batch(self, Mc, q, l, u, z0=None, kind=None, opts=None, want_active=True, out=None):
        """Batched AVI solve; replaces PATHSolver.solve_mcp (src/avi.jl:64-70) per item.

        Mc: (batch, N, N) column-major per item, or (N, N) shared.  q, l, u: (batch, N).
        kind: None | (N,) shared | (batch, N) uint8.  z0 = None is a cold start (z0 = 0, handled in
        the kernel: no reset pass).  `out` may carry the dict of a previous call to reuse its
        buffers.  Returns dict(z, status, resid, pivots, active).
        """
        dev = self._mode(Mc, q, l, u, z0, kind)
        self._bind_stream(dev)
        if not dev:
            Mc, q, l, u = (self._host(a, np.float64) for a in (Mc, q, l, u))
            kind = self._host(kind, np.uint8)
        else:
            self._require_dev64(Mc, q, l, u, z0)
        batch, N = q.shape
        strideM = 0 if Mc.ndim == 2 else N * N
        sk = 0 if (kind is None or kind.ndim == 1) else N
        o = opts if opts is not None else self.default_opts()
        if out is not None and out["z"].shape == q.shape:
            z, status, resid, pivots, active = out["z"], out["status"], out["resid"], out["pivots"], out["active"]
        else:
            z = self._alloc(dev, (batch, N), np.float64)
            status = self._alloc(dev, (batch,), np.int32)
            resid = self._alloc(dev, (batch,), np.float64)
            pivots = self._alloc(dev, (batch,), np.int32)
            active = self._alloc(dev, (batch, N), np.uint8) if want_active else None
        if z0 is None:
            if opts is None:
                o.flags |= _lib.AVI_FLAG_COLD_START
            elif dev:
                z.zero_()
            else:
                z[...] = 0.0
        elif dev:
            z.copy_(z0)
        else:
            z[...] = np.asarray(z0, dtype=np.float64)
        rc = self.lib.qpn_solve_avi_batch(self.ctx, batch, N, _ptr(Mc), strideM, _ptr(q), _ptr(l),
                                          _ptr(u), _ptr(kind), sk, _ptr(z), _ptr(status),
                                          _ptr(resid), _ptr(pivots), _ptr(active), C.byref(o),
                                          MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_solve_avi_batch")
        return dict(z=z, status=status, resid=resid, pivots=pivots, active=active)

    def solve_mcp_csc(self, N, colptr, rowval, nzval, q, l, u, z0, opts=None):
        """One box-MCP in Julia's SparseMatrixCSC{Float64,Int32} layout (1-based): the argument
        list of PATHSolver.solve_mcp at src/avi.jl:64.  Returns (status, z, info)."""
        colptr = np.ascontiguousarray(colptr, dtype=np.int32)
        rowval = np.ascontiguousarray(rowval, dtype=np.int32)
        nzval = np.ascontiguousarray(nzval, dtype=np.float64)
        q, l, u = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, l, u))
        z = np.array(z0, dtype=np.float64, copy=True)
        st, res, piv = C.c_int32(0), C.c_double(0), C.c_int32(0)
        o = opts if opts is not None else self.default_opts()
        rc = self.lib.qpn_solve_mcp_csc(self.ctx, int(N), _ptr(colptr), _ptr(rowval), _ptr(nzval),
                                        _ptr(q), _ptr(l), _ptr(u), _ptr(z), C.byref(st),
                                        C.byref(res), C.byref(piv), C.byref(o))
        self._chk(rc, "qpn_solve_mcp_csc")
        return int(st.value), z, dict(resid=res.value, pivots=piv.value)

    # -- (A3) ------------------------------------------------------------------------------
    def check_avi_batch(self, Mc, q, l, u, z, kind=None, tol=1e-6, want_r=True):
        dev = self._mode(Mc, q, l, u, z, kind)
        self._bind_stream(dev)
        if not dev:
            Mc, q, l, u, z = (self._host(a, np.float64) for a in (Mc, q, l, u, z))
            kind = self._host(kind, np.uint8)
        else:
            self._require_dev64(Mc, q, l, u, z)
        batch, N = q.shape
        strideM = 0 if Mc.ndim == 2 else N * N
        sk = 0 if (kind is None or kind.ndim == 1) else N
        degree = self._alloc(dev, (batch,), np.int32)
        r = self._alloc(dev, (batch, N), np.float64) if want_r else None
        rc = self.lib.qpn_check_avi_batch(self.ctx, batch, N, _ptr(Mc), strideM, _ptr(q), _ptr(l),
                                          _ptr(u), _ptr(kind), sk, _ptr(z), float(tol),
                                          _ptr(degree), _ptr(r), MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_check_avi_batch")
        return degree, r

    # -- (A9) ------------------------------------------------------------------------------
    def comp_indices(self, zv, rv, l, u, tol=1e-2, shift=0):
        dev = self._mode(zv, rv, l, u)
        self._bind_stream(dev)
        if not dev:
            zv, rv, l, u = (self._host(a, np.float64) for a in (zv, rv, l, u))
        else:
            self._require_dev64(zv, rv, l, u)
        count = int(np.prod(zv.shape))
        mask = self._alloc(dev, tuple(zv.shape), np.uint8)
        rc = self.lib.qpn_comp_indices(self.ctx, count, _ptr(zv), _ptr(rv), _ptr(l), _ptr(u),
                                       float(tol), int(shift), _ptr(mask),
                                       MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_comp_indices")
        return mask

    # -- (A5+A6) ---------------------------------------------------------------------------
    def assemble_nodes(self, Qc, Rc, qd, Ac, Bc, l, u, w, out=None):
        """Per-node reduced KKT blocks.  Qc (batch,n,n), Rc (batch,p,n), Ac (batch,n,m),
        Bc (batch,p,m): all column-major per item (see ``colmajor``); w (p,) shared or (batch,p).
        `out` may carry the tuple of a previous call to reuse its buffers."""
        dev = self._mode(Qc, Rc, qd, Ac, Bc, l, u, w)
        self._bind_stream(dev)
        if not dev:
            Qc, Rc, qd, Ac, Bc, l, u, w = (self._host(a, np.float64) for a in (Qc, Rc, qd, Ac, Bc, l, u, w))
        else:
            self._require_dev64(Qc, Rc, qd, Ac, Bc, l, u, w)
        batch, n = qd.shape
        m = l.shape[1]
        p = w.shape[-1]
        sw = 0 if w.ndim == 1 else p
        N = n + m
        if out is not None and out[1].shape == (batch, N):
            Mout, qout, lout, uout, kind = out
        else:
            Mout = self._alloc(dev, (batch, N, N), np.float64)
            qout = self._alloc(dev, (batch, N), np.float64)
            lout = self._alloc(dev, (batch, N), np.float64)
            uout = self._alloc(dev, (batch, N), np.float64)
            kind = self._alloc(dev, (batch, N), np.uint8)
        rc = self.lib.qpn_assemble_nodes(self.ctx, batch, n, m, p, _ptr(Qc), _ptr(Rc), _ptr(qd),
                                         _ptr(Ac), _ptr(Bc), _ptr(l), _ptr(u), _ptr(w), sw,
                                         _ptr(Mout), _ptr(qout), _ptr(lout), _ptr(uout), _ptr(kind),
                                         MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_assemble_nodes")
        return Mout, qout, lout, uout, kind

    # -- (F1) local pieces ------------------------------------------------------------------------
    def recipes_from_masks(self, mask, first=0, count=None):
        """all_Ks (src/avi_solutions.jl:200-215) from one solution's active-set masks (uint8 per row of z): recipes number
        first .. first+count-1 of the Cartesian product of the rows' code sets.  Returns (K [count, N] uint8, total)."""
        dev = self._mode(mask)
        self._bind_stream(dev)
        if not dev:
            mask = self._host(mask, np.uint8)
        N = int(mask.shape[0])
        total = C.c_int64(0)
        rc = self.lib.qpn_recipes_from_masks(self.ctx, N, _ptr(mask), 0, 0, None, C.byref(total), MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_recipes_from_masks")
        tot = int(total.value)
        if count is None:
            count = tot - first
        K = self._alloc(dev, (count, N), np.uint8)
        rc = self.lib.qpn_recipes_from_masks(self.ctx, N, _ptr(mask), int(first), int(count), _ptr(K), None,
                                             MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_recipes_from_masks")
        return K, tot

    def local_pieces(self, Qc, Rc, qd, Ac, Bc, l, u, K, node_of=None):
        """local_piece (src/avi_solutions.jl:400-496, before simplify) for recipes K [pieces, n+m] over node records in the
        ABI layout (as solve_nodes); node_of [pieces] int32 names each recipe's node (default: recipe t <-> node t).
        Returns (Ap [pieces, N+p, 2N] column-major per piece, lp, up [pieces, 2N], keep [pieces, 2N] uint8)."""
        dev = self._mode(Qc, Rc, qd, Ac, Bc, l, u, K, node_of)
        self._bind_stream(dev)
        if not dev:
            Qc, Rc, qd, Ac, Bc, l, u = (self._host(a, np.float64) for a in (Qc, Rc, qd, Ac, Bc, l, u))
            K = self._host(K, np.uint8)
            node_of = self._host(node_of, np.int32)
        else:
            self._require_dev64(Qc, Rc, qd, Ac, Bc, l, u)
        nodes, n = qd.shape
        m = l.shape[1]
        p = Rc.shape[1]
        N = n + m
        pieces = int(K.shape[0])
        Ap = self._alloc(dev, (pieces, N + p, 2 * N), np.float64)
        lp = self._alloc(dev, (pieces, 2 * N), np.float64)
        up = self._alloc(dev, (pieces, 2 * N), np.float64)
        keep = self._alloc(dev, (pieces, 2 * N), np.uint8)
        rc = self.lib.qpn_local_pieces(self.ctx, pieces, nodes, n, m, p, _ptr(Qc), _ptr(Rc), _ptr(qd), _ptr(Ac), _ptr(Bc), _ptr(l),
                                       _ptr(u), _ptr(node_of), _ptr(K), _ptr(Ap), _ptr(lp), _ptr(up), _ptr(keep),
                                       MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_local_pieces")
        return Ap, lp, up, keep

    # -- (F1, a level at a time) --------------------------------------------------------------------
    def recipes_batch(self, masks, offsets):
        """all_Ks (src/avi_solutions.jl:200-215) for MANY solutions in one launch (qpn_recipes_batch): masks [nodes, N] uint8,
        offsets [nodes + 1] int64 (host; node b gets the first offsets[b+1] - offsets[b] recipes of its product).
        Returns (K [total, N] uint8, node_of [total] int32)."""
        dev = self._mode(masks)
        self._bind_stream(dev)
        if not dev:
            masks = self._host(masks, np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        nodes, N = int(masks.shape[0]), int(masks.shape[1])
        if offsets.shape != (nodes + 1,):
            raise ValueError("recipes_batch: offsets must have nodes + 1 entries")
        total = int(offsets[-1])
        K = self._alloc(dev, (total, N), np.uint8)
        node_of = self._alloc(dev, (total,), np.int32)
        rc = self.lib.qpn_recipes_batch(self.ctx, nodes, N, _ptr(masks), _ptr(offsets), _ptr(K), _ptr(node_of),
                                        MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_recipes_batch")
        return K, node_of

    def reduced_pieces(self, Qc, Rc, qd, Ac, Bc, l, u, K, node_of=None, tol=1e-9):
        """local_piece (src/avi_solutions.jl:400-496) for recipes K over node records, with the m multiplier columns eliminated
        through each piece's own equality rows (qpn_reduced_pieces).  Returns (Ar [pieces, n+p, cap] column-major per piece --
        Ar[t].T is the cap x (n+p) row matrix over [x_d; x_p] --, lr, ur [pieces, cap], rows [pieces], flags [pieces]),
        cap = n + 2m."""
        dev = self._mode(Qc, Rc, qd, Ac, Bc, l, u, K, node_of)
        self._bind_stream(dev)
        if not dev:
            Qc, Rc, qd, Ac, Bc, l, u = (self._host(a, np.float64) for a in (Qc, Rc, qd, Ac, Bc, l, u))
            K = self._host(K, np.uint8)
            node_of = self._host(node_of, np.int32)
        else:
            self._require_dev64(Qc, Rc, qd, Ac, Bc, l, u)
        nodes, n = qd.shape
        m = l.shape[1]
        p = Rc.shape[1]
        pieces = int(K.shape[0])
        cap = n + 2 * m
        Ar = self._alloc(dev, (pieces, n + p, cap), np.float64)
        lr = self._alloc(dev, (pieces, cap), np.float64)
        ur = self._alloc(dev, (pieces, cap), np.float64)
        rows = self._alloc(dev, (pieces,), np.int32)
        flags = self._alloc(dev, (pieces,), np.int32)
        rc = self.lib.qpn_reduced_pieces(self.ctx, pieces, nodes, n, m, p, _ptr(Qc), _ptr(Rc), _ptr(qd), _ptr(Ac), _ptr(Bc), _ptr(l),
                                         _ptr(u), _ptr(node_of), _ptr(K), float(tol), _ptr(Ar), _ptr(lr), _ptr(ur), _ptr(rows),
                                         _ptr(flags), MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_reduced_pieces")
        return Ar, lr, ur, rows, flags

    # -- (A6) pool assembly ----------------------------------------------------------------------
    def assemble_pools(self, n_i, m_i, dpos, nd, Qd, Qp, qd, Ad, Bp, l, u, w, form="reduced", share_M=None):
        """combine_gavis (src/avi.jl:305-377) for `batch` instances of one pool shape, on the device.

        Shape: n_i, m_i (per player, pool order), dpos (decision position of every stacked player row), nd.
        Blocks in the ABI layout (column-major), each either shared ((cols, rows)) or per item ((batch, cols, rows)):
        Qd (nd, sn), Qp (p, sn), qd (sn,), Ad (nd, sm), Bp (p, sm), l, u (sm,), w (p,).  form: "reduced" | "reference".
        share_M (default: automatically when Qd and Ad are shared): write ONE M for the whole batch.
        Returns (Mc, q, lo, hi, kind) ready for solve_avi_batch (Mc (N, N) when shared)."""
        from ._lib import POOL_REDUCED, POOL_REFERENCE, PoolShape
        dev = self._mode(Qd, Qp, qd, Ad, Bp, l, u, w)
        self._bind_stream(dev)
        n_i = np.ascontiguousarray(n_i, dtype=np.int32); m_i = np.ascontiguousarray(m_i, dtype=np.int32)
        dpos = np.ascontiguousarray(dpos, dtype=np.int32)
        sn, sm = int(n_i.sum()), int(m_i.sum())
        if not dev:
            Qd, Qp, qd, Ad, Bp, l, u, w = (self._host(a, np.float64) for a in (Qd, Qp, qd, Ad, Bp, l, u, w))
        else:
            self._require_dev64(Qd, Qp, qd, Ad, Bp, l, u, w)
        p = int(w.shape[-1])
        item_dims = dict(Qd=2, Qp=2, qd=1, Ad=2, Bp=2, l=1, u=1, w=1)
        arrs = dict(Qd=Qd, Qp=Qp, qd=qd, Ad=Ad, Bp=Bp, l=l, u=u, w=w)
        sizes = dict(Qd=sn * nd, Qp=sn * p, qd=sn, Ad=sm * nd, Bp=sm * p, l=sm, u=sm, w=p)
        batch = 1
        strides = {}
        for k, a_ in arrs.items():
            if a_.ndim == item_dims[k] + 1:
                batch = max(batch, int(a_.shape[0])); strides[k] = sizes[k]
            elif a_.ndim == item_dims[k]:
                strides[k] = 0
            else:
                raise ValueError(f"assemble_pools: {k} has {a_.ndim} dimensions")
            n_el = int(np.prod(a_.shape[-item_dims[k]:])) if item_dims[k] else 1
            if n_el != sizes[k]:
                raise ValueError(f"assemble_pools: {k} has {n_el} entries per item, the shape says {sizes[k]}")
        if strides["l"] != strides["u"]:
            raise ValueError("assemble_pools: l and u must both be shared or both per item")
        for k, a_ in arrs.items():
            if strides[k] and a_.shape[0] != batch:
                raise ValueError(f"assemble_pools: {k} has batch {a_.shape[0]}, others {batch}")
        fcode = {"reduced": POOL_REDUCED, "reference": POOL_REFERENCE}[form]
        shape = PoolShape(len(n_i), int(nd), p, n_i.ctypes.data, m_i.ctypes.data, dpos.ctypes.data)
        Nn = C.c_int32(0)
        if self.lib.qpn_pool_size(C.byref(shape), fcode, C.byref(Nn)) != 0:
            raise QpnError("qpn_pool_size: bad pool shape")
        N = int(Nn.value)
        if share_M is None:
            share_M = strides["Qd"] == 0 and strides["Ad"] == 0
        Mout = self._alloc(dev, (N, N) if share_M else (batch, N, N), np.float64)
        qout = self._alloc(dev, (batch, N), np.float64)
        lout = self._alloc(dev, (batch, N), np.float64)
        uout = self._alloc(dev, (batch, N), np.float64)
        kind = self._alloc(dev, (batch, N), np.uint8)
        rc = self.lib.qpn_assemble_pools(self.ctx, C.byref(shape), fcode, batch, _ptr(Qd), strides["Qd"], _ptr(Qp), strides["Qp"],
                                         _ptr(qd), strides["qd"], _ptr(Ad), strides["Ad"], _ptr(Bp), strides["Bp"], _ptr(l), _ptr(u),
                                         strides["l"], _ptr(w), strides["w"], _ptr(Mout), 0 if share_M else N * N, _ptr(qout),
                                         _ptr(lout), _ptr(uout), _ptr(kind), MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_assemble_pools")
        return Mout, qout, lout, uout, kind

    # -- (A5+A6+A2+A3+A9 fused) --------------------------------------------------------------
    def solve_nodes(self, Qc, Rc, qd, Ac, Bc, l, u, w, z0=None, opts=None, want_active=True, out=None,
                    x_out=None):
        """Assemble every node's KKT blocks on the fly and solve (one kernel for n, m <= 32); same
        results as assemble_nodes + solve_avi_batch without materialising M.  z = [x_d; lambda].
        x_out (optional, [batch, >= n] fp64, rows may be strided): the primal blocks are also written
        there by the solve itself -- the outer sweep's x[decision_inds] = x_opt[decision_inds]
        (src/algorithm.jl:97-101)."""
        dev = self._mode(Qc, Rc, qd, Ac, Bc, l, u, w, z0)
        self._bind_stream(dev)
        if not dev:
            Qc, Rc, qd, Ac, Bc, l, u, w = (self._host(a, np.float64) for a in (Qc, Rc, qd, Ac, Bc, l, u, w))
        else:
            self._require_dev64(Qc, Rc, qd, Ac, Bc, l, u, w)
        batch, n = qd.shape
        m = l.shape[1]
        p = w.shape[-1]
        sw = 0 if w.ndim == 1 else p
        N = n + m
        o = opts if opts is not None else self.default_opts()
        if out is not None and out["z"].shape == (batch, N):
            z, status, resid, pivots, active = out["z"], out["status"], out["resid"], out["pivots"], out["active"]
        else:
            z = self._alloc(dev, (batch, N), np.float64)
            status = self._alloc(dev, (batch,), np.int32)
            resid = self._alloc(dev, (batch,), np.float64)
            pivots = self._alloc(dev, (batch,), np.int32)
            active = self._alloc(dev, (batch, N), np.uint8) if want_active else None
        if z0 is None:
            if opts is None:
                o.flags |= _lib.AVI_FLAG_COLD_START
            elif dev:
                z.zero_()
            else:
                z[...] = 0.0
        elif dev:
            z.copy_(z0)
        else:
            z[...] = np.asarray(z0, dtype=np.float64)
        sx = self._x_stride(x_out, dev, batch, n)
        rc = self.lib.qpn_solve_nodes_into(self.ctx, batch, n, m, p, _ptr(Qc), _ptr(Rc), _ptr(qd), _ptr(Ac),
                                           _ptr(Bc), _ptr(l), _ptr(u), _ptr(w), sw, _ptr(z), _ptr(status),
                                           _ptr(resid), _ptr(pivots), _ptr(active), C.byref(o),
                                           MEM_DEVICE if dev else MEM_HOST, _ptr(x_out), sx)
        self._chk(rc, "qpn_solve_nodes_into")
        return dict(z=z, status=status, resid=resid, pivots=pivots, active=active)

    def order_nodes_by_pivots(self, pivots):
        """Schedule hint for later solve_nodes calls over the SAME nodes: longest solves first, from the
        pivot counts of an earlier sweep (device or host int32 array).  Results do not depend on it."""
        dev = self._mode(pivots)
        self._bind_stream(dev)
        if not dev:
            pivots = self._host(pivots, np.int32)
        rc = self.lib.qpn_order_nodes_by_pivots(self.ctx, _ptr(pivots), int(pivots.shape[0]),
                                                MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_order_nodes_by_pivots")

    def set_auto_schedule(self, period=16):
        """Period (in calls) of the context's own longest-first schedule refresh for solve_nodes batches that fill the
        GPU; 0 switches it off.  An explicit hint (order_nodes_by_pivots / set_node_order) takes precedence."""
        self._chk(self.lib.qpn_ctx_set_auto_schedule(self.ctx, int(period)), "qpn_ctx_set_auto_schedule")

    def set_option(self, option, value):
        """Per-context route option (include/qpn_hip.h: QPN_OPT_*), e.g. set_option(OPT_MID_ROUTE, 2)."""
        self._chk(self.lib.qpn_ctx_set_option(self.ctx, int(option), int(value)), "qpn_ctx_set_option")

    def set_node_order(self, order=None):
        """Install a caller-made permutation of the nodes as the schedule (None clears the hint)."""
        if order is None:
            self._chk(self.lib.qpn_set_node_order(self.ctx, None, 0, MEM_HOST), "qpn_set_node_order")
            return
        dev = self._mode(order)
        self._bind_stream(dev)
        if not dev:
            order = self._host(order, np.int32)
        rc = self.lib.qpn_set_node_order(self.ctx, _ptr(order), int(order.shape[0]), MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_set_node_order")

    # -- (A8) ------------------------------------------------------------------------------
    # -- multi-GPU: shared iterate buffers, replicas, per-sweep status (include/qpn_hip.h) ----------
    def shared_alloc(self, nbytes, fine_grained=False):
        """Zeroed device buffer + its IPC handle: (address, handle bytes)."""
        from ._lib import IPC_HANDLE_BYTES, SHARED_FINE_GRAINED
        ptr = C.c_void_p()
        h = (C.c_uint8 * IPC_HANDLE_BYTES)()
        rc = self.lib.qpn_shared_alloc(self.ctx, int(nbytes), SHARED_FINE_GRAINED if fine_grained else 0,
                                       C.byref(ptr), h)
        self._chk(rc, "qpn_shared_alloc")
        return int(ptr.value), bytes(h)

    def shared_open(self, handle: bytes) -> int:
        """Map a peer's shared buffer into this process; returns its address here."""
        ptr = C.c_void_p()
        h = (C.c_uint8 * len(handle)).from_buffer_copy(handle)
        self._chk(self.lib.qpn_shared_open(self.ctx, h, C.byref(ptr)), "qpn_shared_open")
        return int(ptr.value)

    def shared_close(self, addr: int):
        self._chk(self.lib.qpn_shared_close(self.ctx, C.c_void_p(addr)), "qpn_shared_close")

    def shared_free(self, addr: int):
        self._chk(self.lib.qpn_shared_free(self.ctx, C.c_void_p(addr)), "qpn_shared_free")

    def set_primal_mirrors(self, own_addr=0, nbytes=0, peer_addrs=()):
        """Later solve_nodes(x_out=...) calls whose x_out lies inside [own_addr, own_addr + nbytes) also store
        every primal block at the same offset of each peer buffer.  No arguments: clear."""
        arr = (C.c_void_p * max(len(peer_addrs), 1))(*[C.c_void_p(a) for a in peer_addrs])
        rc = self.lib.qpn_set_primal_mirrors(self.ctx, C.c_void_p(own_addr), int(nbytes), len(peer_addrs), arr)
        self._chk(rc, "qpn_set_primal_mirrors")

    def sweep_status(self, status, resid, out, rank=0, world=1, boxes=None, epoch=0, timeout_ms=1000):
        """out[0:3] (device fp64, 4 entries) <- (items not solved, max resid, 1), combined over `world` ranks
        through their mailboxes when world > 1 (also the barrier after the replica stores); a missed barrier gives
        out[2] = 0 and out[3] += 1.  Asynchronous on the stream."""
        self._bind_stream(True)
        arr = None
        if world > 1:
            arr = (C.c_void_p * world)(*[C.c_void_p(a) for a in boxes])
        rc = self.lib.qpn_sweep_status(self.ctx, _ptr(status), _ptr(resid), int(status.shape[0]), _ptr(out),
                                       int(rank), int(world), arr, int(epoch), int(timeout_ms))
        self._chk(rc, "qpn_sweep_status")
        return out

    def verify_nodes(self, Qc, Rc, qd, Ac, Bc, l, u, xd, w, tol=1e-4):
        """Batched verify_solution (src/qp_processing.jl:57-149) -> (solution, lambda, path)."""
        dev = self._mode(Qc, Rc, qd, Ac, Bc, l, u, xd, w)
        self._bind_stream(dev)
        if not dev:
            Qc, Rc, qd, Ac, Bc, l, u, xd, w = (self._host(a, np.float64)
                                               for a in (Qc, Rc, qd, Ac, Bc, l, u, xd, w))
        else:
            self._require_dev64(Qc, Rc, qd, Ac, Bc, l, u, xd, w)
        batch, n = qd.shape
        m = l.shape[1]
        p = w.shape[-1]
        sw = 0 if w.ndim == 1 else p
        sol = self._alloc(dev, (batch,), np.int32)
        path = self._alloc(dev, (batch,), np.int32)
        lam = self._alloc(dev, (batch, max(m, 1)), np.float64)
        rc = self.lib.qpn_verify_nodes(self.ctx, batch, n, m, p, _ptr(Qc), _ptr(Rc), _ptr(qd),
                                       _ptr(Ac), _ptr(Bc), _ptr(l), _ptr(u), _ptr(xd), _ptr(w), sw,
                                       float(tol), _ptr(sol), _ptr(lam), _ptr(path),
                                       MEM_DEVICE if dev else MEM_HOST)
        self._chk(rc, "qpn_verify_nodes")
        return sol, lam[:, :m], path


class Nodes:
    """Resident node records (``qpn_nodes_upload``): the records of a level's single-node pools live in HBM owned by the
    library; a sweep hands over only the parameters ``w`` and the output buffers.  What the outer loop
    (src/algorithm.jl:13-117) does between two sweeps -- new parameters, same nodes -- costs no record traffic, and the
    handle remembers what depends on the records alone (whether any node needs the general kernel; the longest-first
    schedule).  ``solve`` = Engine.solve_nodes, ``verify`` = Engine.verify_nodes with the records in place."""

    FIELDS = dict(Qd=0, R=1, qd=2, Ad=3, B=4, l=5, u=6)

    def __init__(self, eng: "Engine", Qc, Rc, qd, Ac, Bc, l, u):
        self.eng = eng
        dev = eng._mode(Qc, Rc, qd, Ac, Bc, l, u)
        eng._bind_stream(dev)
        if not dev:
            Qc, Rc, qd, Ac, Bc, l, u = (eng._host(a, np.float64) for a in (Qc, Rc, qd, Ac, Bc, l, u))
        else:
            eng._require_dev64(Qc, Rc, qd, Ac, Bc, l, u)
        self.batch, self.n = qd.shape
        self.m = l.shape[1]
        self.p = Rc.shape[1]
        h = C.c_void_p()
        rc = eng.lib.qpn_nodes_upload(eng.ctx, self.batch, self.n, self.m, self.p, _ptr(Qc), _ptr(Rc), _ptr(qd), _ptr(Ac),
                                      _ptr(Bc), _ptr(l), _ptr(u), MEM_DEVICE if dev else MEM_HOST, C.byref(h))
        eng._chk(rc, "qpn_nodes_upload")
        self.h = h
        self._fast = None

    def close(self):
        if getattr(self, "h", None) and getattr(self.eng, "ctx", None):
            self.eng.lib.qpn_nodes_free(self.eng.ctx, self.h)
        self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def update(self, field: str, data):
        """Replace one array of the records (same shape), e.g. the bounds after another child piece was chosen."""
        dev = self.eng._mode(data)
        self.eng._bind_stream(dev)
        if not dev:
            data = self.eng._host(data, np.float64)
        else:
            self.eng._require_dev64(data)
        rc = self.eng.lib.qpn_nodes_update(self.eng.ctx, self.h, self.FIELDS[field], _ptr(data), MEM_DEVICE if dev else MEM_HOST)
        self.eng._chk(rc, "qpn_nodes_update")

    def info(self):
        """dict(decline_state, declined, scheduled, symmetric, sweeps) -- see qpn_nodes_info."""
        a = (C.c_int32 * 4)()
        self.eng._chk(self.eng.lib.qpn_nodes_info(self.eng.ctx, self.h, a), "qpn_nodes_info")
        return dict(decline_state=a[0], declined=a[1], scheduled=bool(a[2] & 1), symmetric=bool(a[2] & 2), sweeps=a[3])

    def set_schedule(self, period=16):
        self.eng._chk(self.eng.lib.qpn_nodes_set_schedule(self.eng.ctx, self.h, int(period)), "qpn_nodes_set_schedule")

    def solve(self, w, opts=None, want=("z", "resid", "pivots", "active"), out=None, x_out=None):
        """One sweep over the resident nodes with parameters w ((p,) shared or (batch, p)); cold duals.  `want` names the
        optional outputs (status always comes back); x_out as in Engine.solve_nodes."""
        eng = self.eng
        # the sweep loop's call (same output buffers as the previous sweep, device parameters): everything but w's address is
        # what it was -- the checks and conversions below were done when these buffers were first seen
        fast = self._fast
        if fast is not None and out is fast[0] and x_out is fast[1] and opts is None and _is_dev(w) and w.dtype is torch.float64 \
                and w.stride(-1) == 1:
            # the cached argument tail holds raw device addresses: it is valid only while every tensor it was built from is
            # the same object at the same address (a swapped or deleted dict entry, or a resized tensor, rebuilds it below),
            # and only for a w of the handle's own shape on the handle's device
            same = all((out.get(k) is t) and (t is None or t.data_ptr() == a) for k, t, a in fast[4]) and \
                (x_out is None or x_out.data_ptr() == fast[5])
            w_ok = w.shape[-1] == self.p and w.device == fast[6] and (w.ndim == 1 or (w.ndim == 2 and w.shape[0] == self.batch))
            if same and w_ok:
                eng._bind_stream(True)
                rc = eng.lib.qpn_solve_nodes_h(eng.ctx, self.h, w.data_ptr(), 0 if w.ndim == 1 else w.stride(0), *fast[2])
                eng._chk(rc, "qpn_solve_nodes_h")
                return out
            self._fast = None
        dev = eng._mode(w, x_out)
        eng._bind_stream(dev)
        if not dev:
            w = eng._host(w, np.float64)
        elif w.dtype != torch.float64 or w.stride(-1) != 1:
            raise QpnError("w must be a float64 tensor with unit inner stride")
        N = self.n + self.m
        if w.shape[-1] != self.p or w.ndim > 2 or (w.ndim == 2 and w.shape[0] != self.batch):
            raise QpnError(f"w must have shape ({self.p},) or ({self.batch}, {self.p})")
        if dev and w.device.index != eng.device:
            raise QpnError("w lives on another device than the engine")
        sw = 0 if w.ndim == 1 else int(w.stride(0) if dev else w.strides[0] // 8)
        o = opts if opts is not None else eng.default_opts()
        o.flags |= _lib.AVI_FLAG_COLD_START
        if out is not None:
            # a caller-supplied set of output buffers is checked once, when it is first seen
            spec = dict(status=((self.batch,), np.int32), z=((self.batch, N), np.float64), resid=((self.batch,), np.float64),
                        pivots=((self.batch,), np.int32), active=((self.batch, N), np.uint8))
            for k, (shape, dt) in spec.items():
                t = out.get(k)
                if t is None:
                    if k == "status":
                        raise QpnError("out['status'] is required")
                    continue
                if _is_dev(t) != dev:
                    raise QpnError(f"out['{k}'] and w must both be host arrays or both device tensors")
                ok = tuple(t.shape) == shape and (t.is_contiguous() and t.dtype == _torch_dt(dt) and t.device.index == eng.device
                                                  if dev else t.flags["C_CONTIGUOUS"] and t.dtype == dt)
                if not ok:
                    raise QpnError(f"out['{k}'] must be a contiguous {np.dtype(dt).name} buffer of shape {shape} on the engine's device")
            for k in ("z", "resid", "pivots", "active"):
                out.setdefault(k, None)
        if out is None:
            out = dict(status=eng._alloc(dev, (self.batch,), np.int32),
                       z=eng._alloc(dev, (self.batch, N), np.float64) if "z" in want else None,
                       resid=eng._alloc(dev, (self.batch,), np.float64) if "resid" in want else None,
                       pivots=eng._alloc(dev, (self.batch,), np.int32) if "pivots" in want else None,
                       active=eng._alloc(dev, (self.batch, N), np.uint8) if "active" in want else None)
        sx = eng._x_stride(x_out, dev, self.batch, self.n)
        tail = (_ptr(out["z"]), _ptr(out["status"]), _ptr(out["resid"]), _ptr(out["pivots"]), _ptr(out["active"]), C.byref(o),
                MEM_DEVICE if dev else MEM_HOST, _ptr(x_out), sx)
        rc = eng.lib.qpn_solve_nodes_h(eng.ctx, self.h, _ptr(w), sw, *tail)
        eng._chk(rc, "qpn_solve_nodes_h")
        if dev and opts is None:
            # (o is kept alive: tail holds a reference to it; the tensors the addresses in `tail` came from are recorded with
            #  those addresses, so that the fast path above can tell when they are no longer what they were)
            snap = tuple((k, out.get(k), None if out.get(k) is None else out[k].data_ptr())
                         for k in ("z", "status", "resid", "pivots", "active"))
            self._fast = (out, x_out, tail, o, snap, None if x_out is None else x_out.data_ptr(), w.device)
        return out

    def verify(self, xd, w, tol=1e-4):
        eng = self.eng
        dev = eng._mode(xd, w)
        eng._bind_stream(dev)
        if not dev:
            xd, w = eng._host(xd, np.float64), eng._host(w, np.float64)
        else:
            eng._require_dev64(xd, w)
        sw = 0 if w.ndim == 1 else self.p
        sol = eng._alloc(dev, (self.batch,), np.int32)
        path = eng._alloc(dev, (self.batch,), np.int32)
        lam = eng._alloc(dev, (self.batch, max(self.m, 1)), np.float64)
        rc = eng.lib.qpn_verify_nodes_h(eng.ctx, self.h, _ptr(xd), _ptr(w), sw, float(tol), _ptr(sol), _ptr(lam), _ptr(path),
                                        MEM_DEVICE if dev else MEM_HOST)
        eng._chk(rc, "qpn_verify_nodes_h")
        return sol, lam[:, :self.m], path


_default = {}


def default_engine(device: int = 0) -> Engine:
    """Process-wide engine per device (created on first use)."""
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]
