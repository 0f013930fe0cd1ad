// qpn_capi.hip -- the extern "C" boundary of libqpn_hip.so (include/qpn_hip.h).
// Host-side staging, argument checking and error mapping only; all arithmetic is in the
// HIP kernels (qpn_avi_solve.hip, qpn_kkt.hip, qpn_verify.hip).  No exceptions cross the ABI.
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "qpn_internal.h"

struct qpn_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    // grow-only device workspace used by the host-pointer paths and multi-kernel entry points
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // schedule hint of qpn_solve_nodes (qpn_order_nodes_by_pivots / qpn_set_node_order): context-owned
    int32_t *order = nullptr;
    int32_t order_count = 0;      // 0 = no hint installed
    int32_t order_cap = 0;
    // automatic schedule hint (qpn_ctx_set_auto_schedule): refreshed from a call's own pivot counts every `period`
    // calls of the same batch size, unless the caller has installed a hint of their own
    int32_t auto_period = 16;
    int32_t auto_calls = 0;
    int32_t auto_batch = 0;
    bool order_user = false;
    // replicas of the iterate on peer GPUs (qpn_set_primal_mirrors)
    const double *mirror_own = nullptr;
    size_t mirror_bytes = 0;
    int32_t mirror_count = 0;
    double *mirror_peer[QPN_MAX_MIRRORS] = {};
    // route of mid-size node records (qpn_ctx_set_option QPN_OPT_MID_ROUTE): 1 the fused kernels (one wavefront per node up to 48,
    // one workgroup per node beyond), 0 the route of the large nodes / the general kernels (the tests' cross-check)
    int32_t mid_route = 1;
    // every route option change bumps this; a resident handle that learned its declines under another epoch asks again
    int32_t route_epoch = 0;
    // QPN_OPT_SYM_ROUTE: 1 = resident records whose Qd blocks are all bitwise symmetric take the kernel variants that use it
    int32_t sym_route = 1;
};

namespace {

int fail_hip(qpn_ctx *ctx, hipError_t e, const char *where)
{
    if (ctx) {
        ctx->last_error = std::string(where) + ": " + hipGetErrorString(e);
    }
    return QPN_ERR_HIP;
}
int fail_arg(qpn_ctx *ctx, const char *msg)
{
    if (ctx) ctx->last_error = msg;
    return QPN_ERR_ARG;
}

#define HIPCHK(ctx, call)                                        \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return fail_hip(ctx, e__, #call); \
    } while (0)

int order_reserve(qpn_ctx *ctx, int32_t count);

// carve `bytes` (256-B aligned) out of the ctx workspace; grows it when needed
struct Carver {
    qpn_ctx *ctx;
    size_t need = 0;
    std::vector<std::pair<void **, size_t>> slots;
    explicit Carver(qpn_ctx *c) : ctx(c) {}
    void add(void **p, size_t bytes)
    {
        size_t off = need;
        need += (bytes + 255) & ~(size_t)255;
        slots.emplace_back(p, off);
    }
    int commit()
    {
        if (need > ctx->ws_bytes) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->ws) HIPCHK(ctx, hipFree(ctx->ws));
            ctx->ws = nullptr; ctx->ws_bytes = 0;
            size_t want = need + need / 4;
            HIPCHK(ctx, hipMalloc(&ctx->ws, want));
            ctx->ws_bytes = want;
        }
        for (auto &s : slots) *s.first = static_cast<char *>(ctx->ws) + s.second;
        return QPN_OK;
    }
};

} // namespace

extern "C" {

int qpn_abi_version(void) { return QPN_ABI_VERSION; }

const char *qpn_strerror(int code)
{
    switch (code) {
    case QPN_OK: return "ok";
    case QPN_ERR_ARG: return "bad argument";
    case QPN_ERR_HIP: return "HIP runtime error";
    case QPN_ERR_NODEVICE: return "no gfx950 device visible";
    case QPN_ERR_SIZE: return "problem size not supported";
    default: return "unknown error";
    }
}

void qpn_avi_default_opts(qpn_avi_opts *o)
{
    if (!o) return;
    o->check_tol = 1e-6;   // src/avi.jl:148
    o->piv_tol = 1e-11;
    o->feas_tol = 1e-12;
    o->comp_tol = 1e-2;    // src/avi_solutions.jl:511
    o->max_pivots = 0;
    o->flags = 0;
}

int qpn_ctx_create(int device_id, qpn_ctx **out)
{
    if (!out) return QPN_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return QPN_ERR_NODEVICE;
    if (device_id < 0 || device_id >= count) return QPN_ERR_ARG;
    qpn_ctx *ctx = new (std::nothrow) qpn_ctx();
    if (!ctx) return QPN_ERR_ARG;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return QPN_ERR_HIP; }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx; return QPN_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    // diagnostic builds (-DQPN_DEV_SWITCHES) only: a preset of the mid-size route, read ONCE here; qpn_ctx_set_option overrides it
    if (const char *e = QPN_DEV_ENV("QPN_NODES_MID")) { if (e[0] >= '0' && e[0] <= '1' && !e[1]) ctx->mid_route = e[0] - '0'; }
    *out = ctx;
    return QPN_OK;
}

int qpn_ctx_set_option(qpn_ctx *ctx, int32_t option, int32_t value)
{
    if (!ctx) return QPN_ERR_ARG;
    switch (option) {
    case QPN_OPT_MID_ROUTE:
        if (value < 0 || value > 1) return fail_arg(ctx, "qpn_ctx_set_option: QPN_OPT_MID_ROUTE takes 0 or 1");
        if (ctx->mid_route != value) ctx->route_epoch++;
        ctx->mid_route = value;
        return QPN_OK;
    case QPN_OPT_BIG_ROUTE:
        if (value != 1) return fail_arg(ctx, "qpn_ctx_set_option: QPN_OPT_BIG_ROUTE takes 1 (round 2's route over an assembled M is gone)");
        return QPN_OK;
    case QPN_OPT_SYM_ROUTE:
        if (value < 0 || value > 1) return fail_arg(ctx, "qpn_ctx_set_option: QPN_OPT_SYM_ROUTE takes 0 or 1");
        if (ctx->sym_route != value) ctx->route_epoch++;
        ctx->sym_route = value;
        return QPN_OK;
    default:
        return fail_arg(ctx, "qpn_ctx_set_option: unknown option");
    }
}

int qpn_ctx_destroy(qpn_ctx *ctx)
{
    if (!ctx) return QPN_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->order) (void)hipFree(ctx->order);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return QPN_OK;
}

int qpn_ctx_set_stream(qpn_ctx *ctx, void *hip_stream)
{
    if (!ctx) return QPN_ERR_ARG;
    hipStream_t ns = static_cast<hipStream_t>(hip_stream);   // NULL = HIP's legacy default stream
    // the workspace is shared by all entry points and carved from offset 0 by each: launches still running on the
    // stream that is being left may be using it
    if (ns != ctx->stream) { HIPCHK(ctx, hipSetDevice(ctx->device)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
    ctx->stream = ns;
    return QPN_OK;
}

int qpn_ctx_use_own_stream(qpn_ctx *ctx)
{
    if (!ctx) return QPN_ERR_ARG;
    if (ctx->stream != ctx->own_stream) { HIPCHK(ctx, hipSetDevice(ctx->device)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
    ctx->stream = ctx->own_stream;
    return QPN_OK;
}

int qpn_ctx_synchronize(qpn_ctx *ctx)
{
    if (!ctx) return QPN_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return QPN_OK;
}

const char *qpn_ctx_last_error(qpn_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

#ifdef QPN_DIAG
// diagnostic builds only: run Stage A of the MFMA Schur kernel and dump S, c, W, h (device pointers)
int qpn_debug_schur_stage_a(qpn_ctx *ctx, int32_t batch, int32_t N, const double *M, const double *q,
                            const double *l, const double *u, const uint8_t *kind, int32_t *status,
                            double *S, double *c, double *W, double *h)
{
    AviBatchArgs a{};
    a.batch = batch; a.N = N; a.M = M; a.strideM = (int64_t)N * N; a.q = q; a.l = l; a.u = u; a.kind = kind;
    a.stride_kind = N; a.status = status;
    if (qpn_launch_avi_solve_schur(a, S, c, W, h, ctx->stream) != hipSuccess) return QPN_ERR_HIP;
    return hipStreamSynchronize(ctx->stream) == hipSuccess ? QPN_OK : QPN_ERR_HIP;
}
#endif

#ifdef QPN_STAMPS
// diagnostic builds only: where the next device-path solve writes its [batch][8] cycle sums
static unsigned long long *g_stamps = nullptr;
int qpn_debug_set_stamps(void *p) { g_stamps = static_cast<unsigned long long *>(p); return 0; }
#endif

// -------------------------------------------------------------------------------------------
int qpn_solve_avi_batch(qpn_ctx *ctx, int32_t batch, int32_t N, const double *M, int64_t strideM,
                        const double *q, const double *l, const double *u, const uint8_t *kind,
                        int64_t stride_kind, double *z, int32_t *status, double *resid,
                        int32_t *pivots, uint8_t *active, const qpn_avi_opts *opts, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (batch < 0 || N <= 0) return fail_arg(ctx, "qpn_solve_avi_batch: batch < 0 or N <= 0");
    if (batch == 0) return QPN_OK;
    if (!M || !q || !l || !u || !z || !status) return fail_arg(ctx, "qpn_solve_avi_batch: null pointer");
    if (strideM != 0 && strideM < (int64_t)N * N) return fail_arg(ctx, "qpn_solve_avi_batch: strideM < N*N");
    if (kind && stride_kind != 0 && stride_kind < N) return fail_arg(ctx, "qpn_solve_avi_batch: stride_kind < N");
    if (N > qpn_avi_max_n()) { ctx->last_error = "qpn_solve_avi_batch: N > 1024 not supported by ABI v1"; return QPN_ERR_SIZE; }
    const bool big = N > 64;   // one workgroup per item, dictionary in a workspace (qpn_avi_big.hip)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    qpn_avi_opts o;
    if (opts) o = *opts; else qpn_avi_default_opts(&o);

    AviBatchArgs a{};
    a.batch = batch; a.N = N; a.strideM = strideM; a.stride_kind = kind ? stride_kind : 0;
    a.check_tol = o.check_tol; a.piv_tol = o.piv_tol; a.feas_tol = o.feas_tol; a.comp_tol = o.comp_tol;
    a.max_pivots = o.max_pivots;
    a.flags = o.flags & 0xFFFF;           // (the upper bits are internal: QPN_AVI_IFLAG_*)

    if (mem == QPN_MEM_DEVICE) {
        a.M = M; a.q = q; a.l = l; a.u = u; a.kind = kind; a.z = z; a.status = status;
        a.resid = resid; a.pivots = pivots; a.active = active;
#ifdef QPN_STAMPS
        a.stamps = g_stamps;
#endif
        if (big) {
            double *wsp;
            Carver cvb(ctx);
            cvb.add((void **)&wsp, qpn_avi_big_workspace_bytes(batch, N));
            int rcb = cvb.commit();
            if (rcb != QPN_OK) return rcb;
            HIPCHK(ctx, qpn_launch_avi_solve_big(a, wsp, ctx->stream));
        } else {
            HIPCHK(ctx, qpn_launch_avi_solve(a, ctx->stream));
        }
        return QPN_OK;
    }
    if (mem != QPN_MEM_HOST) return fail_arg(ctx, "qpn_solve_avi_batch: bad mem kind");

    const size_t bN = (size_t)batch * N;
    const size_t mBytes = sizeof(double) * (strideM ? (size_t)(batch - 1) * strideM + (size_t)N * N : (size_t)N * N);
    const size_t kBytes = kind ? (stride_kind ? (size_t)(batch - 1) * stride_kind + N : (size_t)N) : 0;
    double *dM, *dq, *dl, *du, *dz, *dres; int32_t *dst, *dpv; uint8_t *dk = nullptr, *dact;
    Carver cv(ctx);
    cv.add((void **)&dM, mBytes); cv.add((void **)&dq, bN * 8); cv.add((void **)&dl, bN * 8);
    cv.add((void **)&du, bN * 8); cv.add((void **)&dz, bN * 8); cv.add((void **)&dres, (size_t)batch * 8);
    cv.add((void **)&dst, (size_t)batch * 4); cv.add((void **)&dpv, (size_t)batch * 4);
    cv.add((void **)&dact, bN); cv.add((void **)&dk, kBytes ? kBytes : 1);
    double *wsb = nullptr;
    if (big) cv.add((void **)&wsb, qpn_avi_big_workspace_bytes(batch, N));
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dM, M, mBytes, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dq, q, bN * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dl, l, bN * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(du, u, bN * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dz, z, bN * 8, hipMemcpyHostToDevice, s));
    if (kind) HIPCHK(ctx, hipMemcpyAsync(dk, kind, kBytes, hipMemcpyHostToDevice, s));
    a.M = dM; a.q = dq; a.l = dl; a.u = du; a.kind = kind ? dk : nullptr; a.z = dz; a.status = dst;
    a.resid = dres; a.pivots = dpv; a.active = dact;
    if (big) HIPCHK(ctx, qpn_launch_avi_solve_big(a, wsb, s));
    else HIPCHK(ctx, qpn_launch_avi_solve(a, s));
    HIPCHK(ctx, hipMemcpyAsync(z, dz, bN * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(status, dst, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    if (resid) HIPCHK(ctx, hipMemcpyAsync(resid, dres, (size_t)batch * 8, hipMemcpyDeviceToHost, s));
    if (pivots) HIPCHK(ctx, hipMemcpyAsync(pivots, dpv, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    if (active) HIPCHK(ctx, hipMemcpyAsync(active, dact, bN, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

int qpn_solve_mcp_csc(qpn_ctx *ctx, int32_t N, const int32_t *colptr, const int32_t *rowval,
                      const double *nzval, const double *q, const double *l, const double *u,
                      double *z, int32_t *status, double *resid, int32_t *pivots,
                      const qpn_avi_opts *opts)
{
    if (!ctx) return QPN_ERR_ARG;
    if (N <= 0 || !colptr || !q || !l || !u || !z || !status) return fail_arg(ctx, "qpn_solve_mcp_csc: bad argument");
    const int32_t nnz = colptr[N] - 1;
    if (nnz < 0 || (nnz > 0 && (!rowval || !nzval))) return fail_arg(ctx, "qpn_solve_mcp_csc: bad CSC arrays");
    std::vector<double> dense((size_t)N * N, 0.0);
    for (int32_t j = 0; j < N; ++j) {
        for (int32_t t = colptr[j] - 1; t < colptr[j + 1] - 1; ++t) {
            int32_t i = rowval[t] - 1;
            if (i < 0 || i >= N) return fail_arg(ctx, "qpn_solve_mcp_csc: row index out of range");
            dense[(size_t)j * N + i] += nzval[t];
        }
    }
    return qpn_solve_avi_batch(ctx, 1, N, dense.data(), 0, q, l, u, nullptr, 0, z, status, resid,
                               pivots, nullptr, opts, QPN_MEM_HOST);
}

// -------------------------------------------------------------------------------------------
int qpn_check_avi_batch(qpn_ctx *ctx, int32_t batch, int32_t N, const double *M, int64_t strideM,
                        const double *q, const double *l, const double *u, const uint8_t *kind,
                        int64_t stride_kind, const double *z, double tol, int32_t *degree,
                        double *r, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (batch < 0 || N <= 0) return fail_arg(ctx, "qpn_check_avi_batch: bad sizes");
    if (batch == 0) return QPN_OK;
    if (!M || !q || !l || !u || !z || !degree) return fail_arg(ctx, "qpn_check_avi_batch: null pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_check_avi(batch, N, M, strideM, q, l, u, kind, kind ? stride_kind : 0, z,
                                         tol, degree, r, ctx->stream));
        return QPN_OK;
    }
    const size_t bN = (size_t)batch * N;
    const size_t mBytes = sizeof(double) * (strideM ? (size_t)(batch - 1) * strideM + (size_t)N * N : (size_t)N * N);
    const size_t kBytes = kind ? (stride_kind ? (size_t)(batch - 1) * stride_kind + N : (size_t)N) : 0;
    double *dM, *dq, *dl, *du, *dz, *dr; int32_t *dd; uint8_t *dk;
    Carver cv(ctx);
    cv.add((void **)&dM, mBytes); cv.add((void **)&dq, bN * 8); cv.add((void **)&dl, bN * 8);
    cv.add((void **)&du, bN * 8); cv.add((void **)&dz, bN * 8); cv.add((void **)&dr, bN * 8);
    cv.add((void **)&dd, (size_t)batch * 4); cv.add((void **)&dk, kBytes ? kBytes : 1);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dM, M, mBytes, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dq, q, bN * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dl, l, bN * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(du, u, bN * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dz, z, bN * 8, hipMemcpyHostToDevice, s));
    if (kind) HIPCHK(ctx, hipMemcpyAsync(dk, kind, kBytes, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_check_avi(batch, N, dM, strideM, dq, dl, du, kind ? dk : nullptr,
                                     kind ? stride_kind : 0, dz, tol, dd, dr, s));
    HIPCHK(ctx, hipMemcpyAsync(degree, dd, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    if (r) HIPCHK(ctx, hipMemcpyAsync(r, dr, bN * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

int qpn_comp_indices(qpn_ctx *ctx, int64_t count, const double *zv, const double *rv,
                     const double *l, const double *u, double tol, int32_t shift, uint8_t *mask,
                     int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (count < 0 || (shift != 0 && shift != 4)) return fail_arg(ctx, "qpn_comp_indices: bad count/shift");
    if (count == 0) return QPN_OK;
    if (!zv || !rv || !l || !u || !mask) return fail_arg(ctx, "qpn_comp_indices: null pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_comp_indices(count, zv, rv, l, u, tol, shift, mask, ctx->stream));
        return QPN_OK;
    }
    double *dz, *dr, *dl, *du; uint8_t *dm;
    Carver cv(ctx);
    const size_t nb = (size_t)count * 8;
    cv.add((void **)&dz, nb); cv.add((void **)&dr, nb); cv.add((void **)&dl, nb); cv.add((void **)&du, nb);
    cv.add((void **)&dm, (size_t)count);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dz, zv, nb, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dr, rv, nb, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dl, l, nb, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(du, u, nb, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_comp_indices(count, dz, dr, dl, du, tol, shift, dm, s));
    HIPCHK(ctx, hipMemcpyAsync(mask, dm, (size_t)count, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

namespace {
struct NodeSizes { size_t Q, R, q, A, B, lu, w; };
NodeSizes node_sizes(int32_t batch, int32_t n, int32_t m, int32_t p, int64_t stride_w)
{
    NodeSizes s;
    s.Q = (size_t)batch * n * n * 8; s.R = (size_t)batch * n * p * 8; s.q = (size_t)batch * n * 8;
    s.A = (size_t)batch * m * n * 8; s.B = (size_t)batch * m * p * 8; s.lu = (size_t)batch * m * 8;
    s.w = (stride_w ? (size_t)(batch - 1) * stride_w + p : (size_t)p) * 8;
    return s;
}
} // namespace

int qpn_assemble_nodes(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p,
                       const double *Qd, const double *R, const double *qd, const double *Ad,
                       const double *B, const double *l, const double *u, const double *w,
                       int64_t stride_w, double *Mout, double *qout, double *lout, double *uout,
                       uint8_t *kind_out, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (batch < 0 || n <= 0 || m < 0 || p < 0) return fail_arg(ctx, "qpn_assemble_nodes: bad sizes");
    if (batch == 0) return QPN_OK;
    if (!Qd || !qd || (m > 0 && (!Ad || !l || !u)) || (p > 0 && (!R || !w || (m > 0 && !B))) || !Mout || !qout ||
        !lout || !uout || !kind_out)
        return fail_arg(ctx, "qpn_assemble_nodes: null pointer");
    if (stride_w != 0 && stride_w < p) return fail_arg(ctx, "qpn_assemble_nodes: stride_w < p");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_assemble_nodes(batch, n, m, p, Qd, R, qd, Ad, B, l, u, w, stride_w, Mout,
                                              qout, lout, uout, kind_out, ctx->stream));
        return QPN_OK;
    }
    const NodeSizes sz = node_sizes(batch, n, m, p, stride_w);
    const int N = n + m;
    const size_t bN = (size_t)batch * N;
    double *dQ, *dR, *dq, *dA, *dB, *dl, *du, *dw, *dM, *dqo, *dlo, *duo; uint8_t *dk;
    Carver cv(ctx);
    cv.add((void **)&dQ, sz.Q); cv.add((void **)&dR, sz.R + 8); cv.add((void **)&dq, sz.q);
    cv.add((void **)&dA, sz.A + 8); cv.add((void **)&dB, sz.B + 8); cv.add((void **)&dl, sz.lu + 8);
    cv.add((void **)&du, sz.lu + 8); cv.add((void **)&dw, sz.w + 8);
    cv.add((void **)&dM, bN * N * 8); cv.add((void **)&dqo, bN * 8); cv.add((void **)&dlo, bN * 8);
    cv.add((void **)&duo, bN * 8); cv.add((void **)&dk, bN);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dQ, Qd, sz.Q, hipMemcpyHostToDevice, s));
    if (sz.R) HIPCHK(ctx, hipMemcpyAsync(dR, R, sz.R, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dq, qd, sz.q, hipMemcpyHostToDevice, s));
    if (sz.A) HIPCHK(ctx, hipMemcpyAsync(dA, Ad, sz.A, hipMemcpyHostToDevice, s));
    if (sz.B) HIPCHK(ctx, hipMemcpyAsync(dB, B, sz.B, hipMemcpyHostToDevice, s));
    if (sz.lu) {
        HIPCHK(ctx, hipMemcpyAsync(dl, l, sz.lu, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(du, u, sz.lu, hipMemcpyHostToDevice, s));
    }
    if (p > 0) HIPCHK(ctx, hipMemcpyAsync(dw, w, sz.w, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_assemble_nodes(batch, n, m, p, dQ, dR, dq, dA, dB, dl, du, dw, stride_w, dM,
                                          dqo, dlo, duo, dk, s));
    HIPCHK(ctx, hipMemcpyAsync(Mout, dM, bN * N * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(qout, dqo, bN * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(lout, dlo, bN * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(uout, duo, bN * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(kind_out, dk, bN, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

// ---- (F1) local pieces ------------------------------------------------------------------------------------------
int qpn_recipes_from_masks(qpn_ctx *ctx, int32_t N, const uint8_t *mask, int64_t first, int32_t count, uint8_t *K,
                           int64_t *total, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (N <= 0 || !mask || first < 0 || count < 0 || (count > 0 && !K)) return fail_arg(ctx, "qpn_recipes_from_masks: bad argument");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_recipes_from_masks: bad mem kind");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<uint8_t> hm((size_t)N);
    if (mem == QPN_MEM_HOST) memcpy(hm.data(), mask, (size_t)N);
    else { HIPCHK(ctx, hipMemcpyAsync(hm.data(), mask, (size_t)N, hipMemcpyDeviceToHost, s)); HIPCHK(ctx, hipStreamSynchronize(s)); }
    int64_t tot = 1;
    for (int i = 0; i < N; ++i) {
        const int r = __builtin_popcount(hm[i]);
        if (r > 1) tot = (tot > INT64_MAX / r) ? INT64_MAX : tot * r;
    }
    if (total) *total = tot;
    if (count == 0) return QPN_OK;
    if (first >= tot || (int64_t)count > tot - first) return fail_arg(ctx, "qpn_recipes_from_masks: first + count beyond the number of recipes");
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_recipes(N, mask, first, count, K, s));
        return QPN_OK;
    }
    uint8_t *dm, *dK;
    Carver cv(ctx);
    cv.add((void **)&dm, (size_t)N); cv.add((void **)&dK, (size_t)count * N);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(dm, mask, (size_t)N, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_recipes(N, dm, first, count, dK, s));
    HIPCHK(ctx, hipMemcpyAsync(K, dK, (size_t)count * N, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

int qpn_recipes_batch(qpn_ctx *ctx, int32_t nodes, int32_t N, const uint8_t *masks, const int64_t *offsets, uint8_t *K,
                      int32_t *node_of, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (nodes <= 0 || N <= 0 || !masks || !offsets) return fail_arg(ctx, "qpn_recipes_batch: bad argument");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_recipes_batch: bad mem kind");
    if (offsets[0] != 0) return fail_arg(ctx, "qpn_recipes_batch: offsets[0] must be 0");
    for (int b = 0; b < nodes; ++b)
        if (offsets[b + 1] < offsets[b]) return fail_arg(ctx, "qpn_recipes_batch: offsets must not decrease");
    const int64_t total = offsets[nodes];
    if (total == 0) return QPN_OK;
    if (!K || !node_of) return fail_arg(ctx, "qpn_recipes_batch: null output");
    if (total > INT32_MAX) { ctx->last_error = "qpn_recipes_batch: more than 2^31 - 1 recipes in one call"; return QPN_ERR_SIZE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // a node may ask for at most the number of recipes its masks have (host masks are checked here; device masks are the
    // caller's: a count beyond the product wraps around inside the product, it cannot leave the arrays)
    if (mem == QPN_MEM_HOST)
        for (int b = 0; b < nodes; ++b) {
            int64_t tot = 1;
            for (int i = 0; i < N; ++i) { const int r = __builtin_popcount(masks[(size_t)b * N + i]); if (r > 1) tot = (tot > INT64_MAX / r) ? INT64_MAX : tot * r; }
            if (offsets[b + 1] - offsets[b] > tot) return fail_arg(ctx, "qpn_recipes_batch: a node asks for more recipes than its masks have");
        }
    long long *doff; uint8_t *dm = nullptr, *dK = nullptr; int32_t *dno = nullptr;
    Carver cv(ctx);
    cv.add((void **)&doff, (size_t)(nodes + 1) * 8);
    if (mem == QPN_MEM_HOST) { cv.add((void **)&dm, (size_t)nodes * N); cv.add((void **)&dK, (size_t)total * N); cv.add((void **)&dno, (size_t)total * 4); }
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    // (the offsets go through a synchronous copy: the host array is the caller's and may be pageable)
    HIPCHK(ctx, hipMemcpyAsync(doff, offsets, (size_t)(nodes + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_recipes_batch(nodes, N, masks, doff, total, K, node_of, s));
        return QPN_OK;
    }
    HIPCHK(ctx, hipMemcpyAsync(dm, masks, (size_t)nodes * N, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_recipes_batch(nodes, N, dm, doff, total, dK, dno, s));
    HIPCHK(ctx, hipMemcpyAsync(K, dK, (size_t)total * N, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(node_of, dno, (size_t)total * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

int qpn_reduced_pieces(qpn_ctx *ctx, int32_t pieces, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd,
                       const double *R, const double *qd, const double *Ad, const double *B, const double *l,
                       const double *u, const int32_t *node_of, const uint8_t *K, double tol, double *Ar, double *lr, double *ur,
                       int32_t *rows, int32_t *flags, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (pieces < 0 || nodes <= 0 || n <= 0 || m < 0 || p < 0) return fail_arg(ctx, "qpn_reduced_pieces: bad sizes");
    if (pieces == 0) return QPN_OK;
    if (n + m > 512) { ctx->last_error = "qpn_reduced_pieces: n + m <= 512 in ABI v1"; return QPN_ERR_SIZE; }
    if (!Qd || !qd || (m > 0 && (!Ad || !l || !u)) || (p > 0 && (!R || (m > 0 && !B))) || !K || !Ar || !lr || !ur || !rows || !flags)
        return fail_arg(ctx, "qpn_reduced_pieces: null pointer");
    if (!node_of && nodes < pieces) return fail_arg(ctx, "qpn_reduced_pieces: fewer record sets than pieces and no node_of");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_reduced_pieces: bad mem kind");
    if (!(tol >= 0.0)) return fail_arg(ctx, "qpn_reduced_pieces: bad tolerance");
    if (qpn_reduce_pieces_lds(n, m, p) > 60 * 1024) { ctx->last_error = "qpn_reduced_pieces: too many parameters for one workgroup's LDS"; return QPN_ERR_SIZE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int N = n + m;
    const size_t rws = 2 * (size_t)N, cols = (size_t)N + p, cap = (size_t)n + 2 * (size_t)m, oc = (size_t)n + p;
    const NodeSizes sz = node_sizes(nodes, n, m, p, 0);
    if (mem == QPN_MEM_HOST && node_of)
        for (int t = 0; t < pieces; ++t)
            if (node_of[t] < 0 || node_of[t] >= nodes) return fail_arg(ctx, "qpn_reduced_pieces: node_of outside 0..nodes-1");
    double *dAp, *dlp, *dup; uint8_t *dkeep;
    double *dQ = nullptr, *dR = nullptr, *dq = nullptr, *dA = nullptr, *dB = nullptr, *dl = nullptr, *du = nullptr, *dAr = nullptr, *dlr = nullptr,
           *dur = nullptr;
    int32_t *dno = nullptr, *drows = nullptr, *dflags = nullptr; uint8_t *dK = nullptr;
    Carver cv(ctx);
    cv.add((void **)&dAp, (size_t)pieces * rws * cols * 8); cv.add((void **)&dlp, (size_t)pieces * rws * 8);
    cv.add((void **)&dup, (size_t)pieces * rws * 8); cv.add((void **)&dkeep, (size_t)pieces * rws);
    if (mem == QPN_MEM_HOST) {
        cv.add((void **)&dQ, sz.Q); cv.add((void **)&dR, sz.R + 8); cv.add((void **)&dq, sz.q); cv.add((void **)&dA, sz.A + 8);
        cv.add((void **)&dB, sz.B + 8); cv.add((void **)&dl, sz.lu + 8); cv.add((void **)&du, sz.lu + 8);
        if (node_of) cv.add((void **)&dno, (size_t)pieces * 4);
        cv.add((void **)&dK, (size_t)pieces * N); cv.add((void **)&dAr, (size_t)pieces * oc * cap * 8);
        cv.add((void **)&dlr, (size_t)pieces * cap * 8 + 8); cv.add((void **)&dur, (size_t)pieces * cap * 8 + 8);
        cv.add((void **)&drows, (size_t)pieces * 4); cv.add((void **)&dflags, (size_t)pieces * 4);
    }
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_local_pieces(pieces, nodes, n, m, p, Qd, R, qd, Ad, B, l, u, node_of, K, dAp, dlp, dup, dkeep, s));
        HIPCHK(ctx, qpn_launch_reduce_pieces(pieces, n, m, p, tol, dAp, dlp, dup, dkeep, Ar, lr, ur, rows, flags, s));
        return QPN_OK;
    }
    HIPCHK(ctx, hipMemcpyAsync(dQ, Qd, sz.Q, hipMemcpyHostToDevice, s));
    if (sz.R) HIPCHK(ctx, hipMemcpyAsync(dR, R, sz.R, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dq, qd, sz.q, hipMemcpyHostToDevice, s));
    if (sz.A) HIPCHK(ctx, hipMemcpyAsync(dA, Ad, sz.A, hipMemcpyHostToDevice, s));
    if (sz.B) HIPCHK(ctx, hipMemcpyAsync(dB, B, sz.B, hipMemcpyHostToDevice, s));
    if (sz.lu) { HIPCHK(ctx, hipMemcpyAsync(dl, l, sz.lu, hipMemcpyHostToDevice, s)); HIPCHK(ctx, hipMemcpyAsync(du, u, sz.lu, hipMemcpyHostToDevice, s)); }
    if (node_of) HIPCHK(ctx, hipMemcpyAsync(dno, node_of, (size_t)pieces * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dK, K, (size_t)pieces * N, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_local_pieces(pieces, nodes, n, m, p, dQ, dR, dq, dA, dB, dl, du, dno, dK, dAp, dlp, dup, dkeep, s));
    HIPCHK(ctx, qpn_launch_reduce_pieces(pieces, n, m, p, tol, dAp, dlp, dup, dkeep, dAr, dlr, dur, drows, dflags, s));
    HIPCHK(ctx, hipMemcpyAsync(Ar, dAr, (size_t)pieces * oc * cap * 8, hipMemcpyDeviceToHost, s));
    if (cap) {
        HIPCHK(ctx, hipMemcpyAsync(lr, dlr, (size_t)pieces * cap * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(ur, dur, (size_t)pieces * cap * 8, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipMemcpyAsync(rows, drows, (size_t)pieces * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(flags, dflags, (size_t)pieces * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

int qpn_local_pieces(qpn_ctx *ctx, int32_t pieces, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd,
                     const double *R, const double *qd, const double *Ad, const double *B, const double *l,
                     const double *u, const int32_t *node_of, const uint8_t *K, double *Ap, double *lp, double *up,
                     uint8_t *keep, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (pieces < 0 || nodes <= 0 || n <= 0 || m < 0 || p < 0) return fail_arg(ctx, "qpn_local_pieces: bad sizes");
    if (pieces == 0) return QPN_OK;
    if (n + m > 512) { ctx->last_error = "qpn_local_pieces: n + m <= 512 in ABI v1"; return QPN_ERR_SIZE; }
    if (!Qd || !qd || (m > 0 && (!Ad || !l || !u)) || (p > 0 && (!R || (m > 0 && !B))) || !K || !Ap || !lp || !up || !keep)
        return fail_arg(ctx, "qpn_local_pieces: null pointer");
    if (!node_of && nodes < pieces) return fail_arg(ctx, "qpn_local_pieces: fewer record sets than pieces and no node_of");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_local_pieces: bad mem kind");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (mem == QPN_MEM_DEVICE) {
        HIPCHK(ctx, qpn_launch_local_pieces(pieces, nodes, n, m, p, Qd, R, qd, Ad, B, l, u, node_of, K, Ap, lp, up, keep, s));
        return QPN_OK;
    }
    if (node_of)
        for (int t = 0; t < pieces; ++t)
            if (node_of[t] < 0 || node_of[t] >= nodes) return fail_arg(ctx, "qpn_local_pieces: node_of outside 0..nodes-1");
    const int N = n + m;
    const NodeSizes sz = node_sizes(nodes, n, m, p, 0);
    const size_t rows = 2 * (size_t)N, cols = (size_t)N + p;
    double *dQ, *dR, *dq, *dA, *dB, *dl, *du, *dAp, *dlp, *dup; int32_t *dno = nullptr; uint8_t *dK, *dkeep;
    Carver cv(ctx);
    cv.add((void **)&dQ, sz.Q); cv.add((void **)&dR, sz.R + 8); cv.add((void **)&dq, sz.q); cv.add((void **)&dA, sz.A + 8);
    cv.add((void **)&dB, sz.B + 8); cv.add((void **)&dl, sz.lu + 8); cv.add((void **)&du, sz.lu + 8);
    if (node_of) cv.add((void **)&dno, (size_t)pieces * 4);
    cv.add((void **)&dK, (size_t)pieces * N); cv.add((void **)&dAp, (size_t)pieces * rows * cols * 8);
    cv.add((void **)&dlp, (size_t)pieces * rows * 8); cv.add((void **)&dup, (size_t)pieces * rows * 8);
    cv.add((void **)&dkeep, (size_t)pieces * rows);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(dQ, Qd, sz.Q, hipMemcpyHostToDevice, s));
    if (sz.R) HIPCHK(ctx, hipMemcpyAsync(dR, R, sz.R, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dq, qd, sz.q, hipMemcpyHostToDevice, s));
    if (sz.A) HIPCHK(ctx, hipMemcpyAsync(dA, Ad, sz.A, hipMemcpyHostToDevice, s));
    if (sz.B) HIPCHK(ctx, hipMemcpyAsync(dB, B, sz.B, hipMemcpyHostToDevice, s));
    if (sz.lu) { HIPCHK(ctx, hipMemcpyAsync(dl, l, sz.lu, hipMemcpyHostToDevice, s)); HIPCHK(ctx, hipMemcpyAsync(du, u, sz.lu, hipMemcpyHostToDevice, s)); }
    if (node_of) HIPCHK(ctx, hipMemcpyAsync(dno, node_of, (size_t)pieces * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dK, K, (size_t)pieces * N, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_local_pieces(pieces, nodes, n, m, p, dQ, dR, dq, dA, dB, dl, du, dno, dK, dAp, dlp, dup, dkeep, s));
    HIPCHK(ctx, hipMemcpyAsync(Ap, dAp, (size_t)pieces * rows * cols * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(lp, dlp, (size_t)pieces * rows * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(up, dup, (size_t)pieces * rows * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(keep, dkeep, (size_t)pieces * rows, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

// ---- (A6) pool assembly -------------------------------------------------------------------------------------
int qpn_pool_size(const qpn_pool_shape *sh, int form, int32_t *N)
{
    if (!sh || !N || sh->players <= 0 || sh->nd <= 0 || sh->p < 0 || !sh->n_i || !sh->m_i) return QPN_ERR_ARG;
    if (form != QPN_POOL_REDUCED && form != QPN_POOL_REFERENCE) return QPN_ERR_ARG;
    int64_t sn = 0, sm = 0;
    for (int i = 0; i < sh->players; ++i) {
        if (sh->n_i[i] <= 0 || sh->m_i[i] < 0) return QPN_ERR_ARG;
        sn += sh->n_i[i]; sm += sh->m_i[i];
    }
    const int64_t n = form == QPN_POOL_REFERENCE ? sh->nd + sn + 2 * sm : sh->nd + sm;
    if (n > 1 << 20) return QPN_ERR_SIZE;
    *N = (int32_t)n;
    return QPN_OK;
}

int qpn_assemble_pools(qpn_ctx *ctx, const qpn_pool_shape *sh, int form, int32_t batch, const double *Qd,
                       int64_t stride_Qd, const double *Qp, int64_t stride_Qp, const double *qd, int64_t stride_qd,
                       const double *Ad, int64_t stride_Ad, const double *Bp, int64_t stride_Bp, const double *l,
                       const double *u, int64_t stride_lu, const double *w, int64_t stride_w, double *Mout,
                       int64_t strideM, double *qout, double *lout, double *uout, uint8_t *kind_out, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    int32_t N = 0;
    int rc = qpn_pool_size(sh, form, &N);
    if (rc != QPN_OK) return rc == QPN_ERR_ARG ? fail_arg(ctx, "qpn_assemble_pools: bad pool shape or form") : rc;
    if (batch < 0) return fail_arg(ctx, "qpn_assemble_pools: batch < 0");
    if (batch == 0) return QPN_OK;
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_assemble_pools: bad mem kind");
    if (!sh->dpos) return fail_arg(ctx, "qpn_assemble_pools: null dpos");
    const int nd = sh->nd, p = sh->p;
    int sn = 0, sm = 0;
    for (int i = 0; i < sh->players; ++i) { sn += sh->n_i[i]; sm += sh->m_i[i]; }
    if (!Qd || !qd || (sm > 0 && (!Ad || !l || !u)) || (p > 0 && (!Qp || !w || (sm > 0 && !Bp))) || !Mout || !qout || !lout ||
        !uout || !kind_out)
        return fail_arg(ctx, "qpn_assemble_pools: null pointer");
    // the index maps of the shape
    std::vector<int32_t> maps((size_t)2 * sn + sm + nd, -1);
    int32_t *xi_owner = maps.data(), *xi_dpos = xi_owner + sn, *con_owner = xi_dpos + sn, *dec_src = con_owner + sm;
    {
        int t = 0, k = 0;
        for (int i = 0; i < sh->players; ++i) {
            for (int e = 0; e < sh->n_i[i]; ++e, ++t) {
                const int d = sh->dpos[t];
                if (d < 0 || d >= nd) return fail_arg(ctx, "qpn_assemble_pools: dpos outside 0..nd-1");
                xi_owner[t] = i; xi_dpos[t] = d;
                if (form == QPN_POOL_REDUCED) {
                    if (dec_src[d] >= 0) return fail_arg(ctx, "qpn_assemble_pools: reduced form needs disjoint decision sets");
                    dec_src[d] = t;
                }
            }
            for (int r = 0; r < sh->m_i[i]; ++r, ++k) con_owner[k] = i;
        }
        if (form == QPN_POOL_REDUCED)
            for (int d = 0; d < nd; ++d)
                if (dec_src[d] < 0) return fail_arg(ctx, "qpn_assemble_pools: a decision position belongs to no player");
    }
    const size_t szQd = (size_t)sn * nd, szQp = (size_t)sn * p, szAd = (size_t)sm * nd, szBp = (size_t)sm * p;
    auto bad_stride = [&](int64_t st, size_t need) { return st != 0 && (st < 0 || (size_t)st < need); };
    if (bad_stride(stride_Qd, szQd) || bad_stride(stride_Qp, szQp) || bad_stride(stride_qd, (size_t)sn) || bad_stride(stride_Ad, szAd) ||
        bad_stride(stride_Bp, szBp) || bad_stride(stride_lu, (size_t)sm) || bad_stride(stride_w, (size_t)p) ||
        bad_stride(strideM, (size_t)N * N))
        return fail_arg(ctx, "qpn_assemble_pools: item stride smaller than the item");
    if (strideM == 0 && batch > 1 && (stride_Qd != 0 || stride_Ad != 0))
        return fail_arg(ctx, "qpn_assemble_pools: a shared M needs shared Qd and Ad");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    auto span = [&](int64_t st, size_t item) { return (st ? (size_t)(batch - 1) * (size_t)st + item : item) * 8; };
    const size_t bQd = span(stride_Qd, szQd), bQp = span(stride_Qp, szQp), bqd = span(stride_qd, sn), bAd = span(stride_Ad, szAd),
                 bBp = span(stride_Bp, szBp), blu = span(stride_lu, sm), bw = span(stride_w, p);
    const size_t bM = span(strideM, (size_t)N * N), bN = (size_t)batch * N;
    int32_t *dmaps;
    double *dQd = nullptr, *dQp = nullptr, *dqd = nullptr, *dAd = nullptr, *dBp = nullptr, *dl = nullptr, *du = nullptr, *dw = nullptr;
    double *dM = nullptr, *dq = nullptr, *dlo = nullptr, *dhi = nullptr; uint8_t *dk = nullptr;
    Carver cv(ctx);
    cv.add((void **)&dmaps, maps.size() * 4);
    if (mem == QPN_MEM_HOST) {
        cv.add((void **)&dQd, bQd + 8); cv.add((void **)&dQp, bQp + 8); cv.add((void **)&dqd, bqd + 8); cv.add((void **)&dAd, bAd + 8);
        cv.add((void **)&dBp, bBp + 8); cv.add((void **)&dl, blu + 8); cv.add((void **)&du, blu + 8); cv.add((void **)&dw, bw + 8);
        cv.add((void **)&dM, bM); cv.add((void **)&dq, bN * 8); cv.add((void **)&dlo, bN * 8); cv.add((void **)&dhi, bN * 8);
        cv.add((void **)&dk, bN);
    }
    rc = cv.commit();
    if (rc != QPN_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(dmaps, maps.data(), maps.size() * 4, hipMemcpyHostToDevice, s));
    QpnPoolLaunch L{};
    L.batch = batch; L.form = form; L.nd = nd; L.sn = sn; L.sm = sm; L.p = p;
    L.xi_owner = dmaps; L.xi_dpos = dmaps + sn; L.con_owner = dmaps + 2 * sn; L.dec_src = dmaps + 2 * sn + sm;
    L.s_Qd = stride_Qd; L.s_Qp = stride_Qp; L.s_qd = stride_qd; L.s_Ad = stride_Ad; L.s_Bp = stride_Bp; L.s_lu = stride_lu;
    L.s_w = stride_w; L.s_M = strideM;
    if (mem == QPN_MEM_HOST) {
        HIPCHK(ctx, hipMemcpyAsync(dQd, Qd, bQd, hipMemcpyHostToDevice, s));
        if (szQp) HIPCHK(ctx, hipMemcpyAsync(dQp, Qp, bQp, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(dqd, qd, bqd, hipMemcpyHostToDevice, s));
        if (szAd) HIPCHK(ctx, hipMemcpyAsync(dAd, Ad, bAd, hipMemcpyHostToDevice, s));
        if (szBp) HIPCHK(ctx, hipMemcpyAsync(dBp, Bp, bBp, hipMemcpyHostToDevice, s));
        if (sm) { HIPCHK(ctx, hipMemcpyAsync(dl, l, blu, hipMemcpyHostToDevice, s)); HIPCHK(ctx, hipMemcpyAsync(du, u, blu, hipMemcpyHostToDevice, s)); }
        if (p) HIPCHK(ctx, hipMemcpyAsync(dw, w, bw, hipMemcpyHostToDevice, s));
        L.Qd = dQd; L.Qp = dQp; L.qd = dqd; L.Ad = dAd; L.Bp = dBp; L.l = dl; L.u = du; L.w = dw;
        L.M = dM; L.q = dq; L.lo = dlo; L.hi = dhi; L.kind = dk;
    } else {
        // (the maps were copied from a host vector that dies with this call: the copy must have left it)
        L.Qd = Qd; L.Qp = Qp; L.qd = qd; L.Ad = Ad; L.Bp = Bp; L.l = l; L.u = u; L.w = w;
        L.M = Mout; L.q = qout; L.lo = lout; L.hi = uout; L.kind = kind_out;
    }
    HIPCHK(ctx, qpn_launch_assemble_pools(L, s));
    if (mem == QPN_MEM_HOST) {
        HIPCHK(ctx, hipMemcpyAsync(Mout, dM, bM, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(qout, dq, bN * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(lout, dlo, bN * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(uout, dhi, bN * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(kind_out, dk, bN, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipStreamSynchronize(s));      // host mode: results; device mode: the pageable maps copy must be done
    return QPN_OK;
}

int qpn_solve_nodes(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                    const double *R, const double *qd, const double *Ad, const double *B,
                    const double *l, const double *u, const double *w, int64_t stride_w, double *z,
                    int32_t *status, double *resid, int32_t *pivots, uint8_t *active,
                    const qpn_avi_opts *opts, int mem)
{
    return qpn_solve_nodes_into(ctx, batch, n, m, p, Qd, R, qd, Ad, B, l, u, w, stride_w, z, status, resid,
                                pivots, active, opts, mem, nullptr, 0);
}

} // extern "C"

// resident node records (qpn_nodes_upload): library-owned copies in HBM + what depends on them alone
struct qpn_nodes {
    int device = 0;
    int32_t batch = 0, n = 0, m = 0, p = 0;
    double *buf = nullptr;                      // one allocation, the seven arrays carved from it
    double *f[7] = {};                          // QPN_NODE_QD .. QPN_NODE_U
    size_t fbytes[7] = {};
    // does any node of these records need the general kernel?  0 unknown, 1 a count is on its way to the host,
    // 2 none (sweeps are ONE launch), 3 some
    int decl_state = 0;
    int32_t *decl_dev = nullptr;                // device counter the fused kernel adds to
    int32_t *decl_host = nullptr;               // pinned host copy
    hipEvent_t decl_ev = nullptr;
    // longest-first schedule of these nodes
    int32_t *order = nullptr;
    int32_t *key = nullptr;                     // smoothed pivot counts the order is made from
    bool order_valid = false;
    int32_t period = 16, calls = 0;
    // every Qd block bitwise symmetric?  Settled when the records arrive (one pass, nodes_check_symmetry)
    bool sym = false;
    // the context's route epoch the decline knowledge was learned under (qpn_ctx_set_option bumps it: another kernel variant
    // applies its pivot test to slightly different numbers, so "no node declines" has to be asked again)
    int32_t route_epoch = 0;
};

namespace {

struct NodeDev {                // device views of one qpn_solve_nodes call
    const double *Q, *R, *q, *A, *B, *l, *u, *w;
    double *z; int32_t *st; double *res; int32_t *pv; uint8_t *act;
};

// one pass over the resident Qd blocks (behind the copies on the context's stream); waits for the answer
hipError_t nodes_check_symmetry(qpn_ctx *ctx, qpn_nodes *h)
{
    h->sym = false;
    hipError_t e = hipMemsetAsync(h->decl_dev, 0, 4, ctx->stream);       // (the decline counter is idle: no solve is in flight)
    if (e == hipSuccess) e = qpn_launch_qd_asymmetry(h->batch, h->n, h->f[QPN_NODE_QD], h->decl_dev, ctx->stream);
    int32_t asym = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(&asym, h->decl_dev, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) h->sym = asym == 0;
    return e;
}

void nodes_poll_declines(qpn_nodes *h)
{
    if (h && h->decl_state == 1 && hipEventQuery(h->decl_ev) == hipSuccess) h->decl_state = (*h->decl_host == 0) ? 2 : 3;
}
// a route option changed since the handle last learned its declines: forget the answer (after the count in flight has landed)
void nodes_sync_route(qpn_ctx *ctx, qpn_nodes *h)
{
    if (!h || h->route_epoch == ctx->route_epoch) return;
    if (h->decl_state == 1) (void)hipEventSynchronize(h->decl_ev);
    h->decl_state = 0;
    h->route_epoch = ctx->route_epoch;
}

// The launches of one sweep over device-resident records and outputs.  `h` (may be null) owns the records:
// its decline knowledge and its schedule are used and refreshed.  wM.. = assembled-block workspace for the
// general path (null only when h knows that no node declines).
int solve_nodes_launch(qpn_ctx *ctx, qpn_nodes *h, int32_t batch, int32_t n, int32_t m, int32_t p, const NodeDev &d,
                       int64_t stride_w, const qpn_avi_opts &o, double *x_dev, int64_t stride_x, double *wM, double *wq,
                       double *wl, double *wu, uint8_t *wk, double *wbig, bool wg_shape)
{
    hipStream_t s = ctx->stream;
    const int N = n + m;
    AviBatchArgs a{};
    a.batch = batch; a.N = N; a.z = d.z; a.status = d.st; a.resid = d.res; a.pivots = d.pv; a.active = d.act;
    a.check_tol = o.check_tol; a.piv_tol = o.piv_tol; a.feas_tol = o.feas_tol; a.comp_tol = o.comp_tol;
    a.max_pivots = o.max_pivots; a.flags = o.flags & 0xFFFF;
    a.nd = NodeSrc{n, m, p, d.Q, d.R, d.q, d.A, d.B, d.l, d.u, d.w, stride_w, (h && h->sym && ctx->sym_route == 1) ? 1 : 0};
#ifdef QPN_STAMPS
    a.stamps = g_stamps;
#endif
    const bool mfma_shape = n <= 32 && m <= 32 && m >= 1;
    bool x_in_kernel = false;
    if (x_dev && mfma_shape) { a.x = x_dev; a.stride_x = stride_x; }      // written by the solve kernels themselves
    // replicas on peer GPUs: only when the whole written range lies inside the registered buffer
    const size_t x_span = x_dev ? ((size_t)(batch - 1) * (size_t)stride_x + (size_t)n) * 8 : 0;
    const bool mirrored = x_dev && ctx->mirror_count > 0 && (const char *)x_dev >= (const char *)ctx->mirror_own &&
                          (const char *)x_dev + x_span <= (const char *)ctx->mirror_own + ctx->mirror_bytes;
    const ptrdiff_t x_off = mirrored ? x_dev - ctx->mirror_own : 0;
    if (mirrored && mfma_shape) {
        a.n_mirror = ctx->mirror_count;
        for (int k = 0; k < ctx->mirror_count; ++k) a.mirror[k] = ctx->mirror_peer[k] + x_off;
    }
    if (mfma_shape) {
        // schedule hint (longest first): the handle's own, else the context's
        // the smoothed counts are fed by every sweep while the schedule settles (128 sweeps), by every fourth one afterwards:
        // a sample of the sweeps tells the order as well, and the solve kernel's read-modify-write of its node's key is
        // 0.5 % of the sweep
        if (h && h->period > 0 && batch > 4096 && (h->calls < 128 || (h->calls & 3) == 0)) a.sched_key = h->key;
        if (h && h->order_valid) a.order = h->order;
        else if (ctx->order_count == batch && (!h || ctx->order_user)) a.order = ctx->order;     // a caller-installed order also serves handles
        bool need_general = true;
        if (h) {
            nodes_poll_declines(h);
            need_general = h->decl_state != 2;
            if (h->decl_state == 0) {
                HIPCHK(ctx, hipMemsetAsync(h->decl_dev, 0, 4, s));
                a.decl_count = h->decl_dev;
            }
        }
        // fused kernel; items it declines (status = -1) are assembled and solved by the general kernel
        // (one small scan-mode launch: its waves pick the flagged items, assemble their blocks into the
        // workspace and solve them)
        HIPCHK(ctx, qpn_launch_avi_solve_schur_nodes(a, s));
        if (need_general) {
            if (!wM) return fail_arg(ctx, "qpn_solve_nodes: internal error (no workspace for the general path)");
            AviBatchArgs g = a;
            g.decl_count = nullptr;
            g.M = wM; g.strideM = (int64_t)N * N; g.q = wq; g.l = wl; g.u = wu; g.kind = wk; g.stride_kind = N;
            g.only_if = d.st; g.only_if_value = -1; g.scan = 1; g.assemble_first = 1;
            HIPCHK(ctx, qpn_launch_avi_solve_reg(g, s));
        }
        if (h && h->decl_state == 0) {
            // which nodes decline depends on Qd, Ad, l, u alone (block pivots of H, equality rows), never on w: ask once
            HIPCHK(ctx, hipMemcpyAsync(h->decl_host, h->decl_dev, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(ctx, hipEventRecord(h->decl_ev, s));
            h->decl_state = 1;
        }
        if (h) {
            // the handle's own longest-first schedule for the NEXT sweeps (launches that fill the GPU only)
            if (h->period > 0 && batch > 4096) {
                // the solve kernel keeps the smoothed counts up to date (every sweep); the order is re-sorted from them every
                // `period` sweeps while they settle (eight refreshes), every 4 x period afterwards: the sort is a
                // one-workgroup launch (14 us) that the next sweep waits for
                const int32_t per = h->calls < 8 * h->period ? h->period : 4 * h->period;
                if (h->calls % per == 0) {
                    HIPCHK(ctx, qpn_launch_order_by_pivots(nullptr, batch, h->order, s, h->key));
                    h->order_valid = true;
                }
                h->calls++;
            }
        } else if (ctx->auto_period > 0 && !ctx->order_user && d.pv && batch > 4096) {
            // automatic longest-first schedule for the NEXT calls over this batch (launches that fill the GPU only)
            if (ctx->auto_batch != batch) { ctx->auto_batch = batch; ctx->auto_calls = 0; }
            if (ctx->auto_calls % ctx->auto_period == 0) {
                int rc = order_reserve(ctx, batch);
                if (rc != QPN_OK) return rc;
                HIPCHK(ctx, qpn_launch_order_by_pivots(d.pv, batch, ctx->order, s));
                ctx->order_count = batch;
            }
            ctx->auto_calls++;
        }
    } else if (wg_shape) {
        // mid-size nodes (n, m <= 128): one wavefront (max(n, m) <= 48) or one workgroup per node straight from the records, ONE
        // launch (qpn_avi_schur48.hip, qpn_avi_schur_wg.hip, qpn_avi_schur_wg2.hip); what they decline (status = -1) is assembled
        // and solved by the general kernels in gated launches
        bool need_general = true;
        if (h) {
            nodes_poll_declines(h);
            need_general = h->decl_state != 2;
            if (h->decl_state == 0) {
                HIPCHK(ctx, hipMemsetAsync(h->decl_dev, 0, 4, s));
                a.decl_count = h->decl_dev;
            }
        }
        {
            // the kernel writes the primal blocks into the iterate itself once it is known that nothing declines (the
            // general kernels behind it do not); until then the strided copy below does
            if (x_dev && !need_general) {
                a.x = x_dev; a.stride_x = stride_x; x_in_kernel = true;
                if (mirrored) {
                    a.n_mirror = ctx->mirror_count;
                    for (int k = 0; k < ctx->mirror_count; ++k) a.mirror[k] = ctx->mirror_peer[k] + x_off;
                }
            }
            // the handle's longest-first schedule as in the 32-class (launches beyond the resident set only: 2 048 wavefronts of the
            // one-wavefront kernel, 1 024 workgroups of the 49-64 class, 256 of the 65-128 class): the kernel feeds the smoothed
            // pivot counts, the order is re-sorted from them
            const bool one_wave = qpn_schur48_shape(n, m), two_role = qpn_schur_wg2_shape(n, m);
            const bool sched = h && h->period > 0 && batch > (one_wave ? 2048 : (two_role ? 256 : 1024));
            if (sched) { a.sched_key = h->key; if (h->order_valid) a.order = h->order; }
            if (two_role) HIPCHK(ctx, qpn_launch_schur_wg2_nodes(a, s));
            else if (one_wave) HIPCHK(ctx, qpn_launch_avi_solve_schur48_nodes(a, s));
            else HIPCHK(ctx, qpn_launch_schur_wg_nodes(a, s));
            a.sched_key = nullptr; a.order = nullptr;
            if (sched) {
                const int32_t per = h->calls < 8 * h->period ? h->period : 4 * h->period;
                if (h->calls % per == 0) {
                    HIPCHK(ctx, qpn_launch_order_by_pivots(nullptr, batch, h->order, s, h->key));
                    h->order_valid = true;
                }
                h->calls++;
            }
            a.x = nullptr; a.n_mirror = 0;
        }
        if (need_general) {
            if (!wM) return fail_arg(ctx, "qpn_solve_nodes: internal error (no workspace for the general path)");
            HIPCHK(ctx, qpn_launch_assemble_nodes(batch, n, m, p, d.Q, d.R, d.q, d.A, d.B, d.l, d.u, d.w, stride_w, wM, wq, wl,
                                                  wu, wk, s, d.st, -1));
            AviBatchArgs g = a;
            g.decl_count = nullptr;
            g.M = wM; g.strideM = (int64_t)N * N; g.q = wq; g.l = wl; g.u = wu; g.kind = wk; g.stride_kind = N;
            g.only_if = d.st; g.only_if_value = -1;
            if (N > 64) HIPCHK(ctx, qpn_launch_avi_solve_big(g, wbig, s));
            else HIPCHK(ctx, qpn_launch_avi_solve(g, s));
        }
        if (h && h->decl_state == 0) {
            HIPCHK(ctx, hipMemcpyAsync(h->decl_host, h->decl_dev, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(ctx, hipEventRecord(h->decl_ev, s));
            h->decl_state = 1;
        }
    } else if (qpn_schur_big2_shape(n, m) && wbig && (o.max_pivots <= 0 || o.max_pivots - n >= 1)) {
        // large nodes (BASELINE config 5): the blocked crash straight from the records (qpn_avi_schur_big2.hip), the
        // delayed-update Lemke kernel on the Schur problems, read-back and post-check on the records; no M is assembled unless a
        // node declines -- those (status = -1) are assembled and solved by the general kernel in gated launches
        SchurBigWs sw{};
        double *dict = wbig;
        void *sb = wbig + (size_t)batch * (size_t)N * (size_t)(N + 1);
        HIPCHK(ctx, qpn_launch_schur_big2_stage_a(a, sb, dict, &sw, s));
        // Stage B: resident records with symmetric Qd blocks (a.nd.sym) go through block principal pivoting first; whatever it
        // leaves (and every node otherwise) is the delayed-update Lemke kernel's
        // (a caller-set pivot budget is Lemke's to count: its pivots are the unit of max_pivots)
        const bool bpp = a.nd.sym && o.max_pivots <= 0;
        if (bpp) HIPCHK(ctx, qpn_launch_schur_big_bpp(a, sw, dict, s));
        HIPCHK(ctx, qpn_launch_schur_big_lemke(a, sw, dict, s, bpp ? 1 : 0));
        HIPCHK(ctx, qpn_launch_schur_big2_finish(a, sw, s));
        HIPCHK(ctx, qpn_launch_assemble_nodes(batch, n, m, p, d.Q, d.R, d.q, d.A, d.B, d.l, d.u, d.w, stride_w, wM, wq, wl,
                                              wu, wk, s, d.st, -1));
        AviBatchArgs g = a;
        g.M = wM; g.strideM = (int64_t)N * N; g.q = wq; g.l = wl; g.u = wu; g.kind = wk; g.stride_kind = N;
        g.only_if = d.st; g.only_if_value = -1;
        HIPCHK(ctx, qpn_launch_avi_solve_big(g, wbig, s));
    } else {
        HIPCHK(ctx, qpn_launch_assemble_nodes(batch, n, m, p, d.Q, d.R, d.q, d.A, d.B, d.l, d.u, d.w, stride_w, wM, wq, wl,
                                              wu, wk, s));
        AviBatchArgs g = a;
        g.M = wM; g.strideM = (int64_t)N * N; g.q = wq; g.l = wl; g.u = wu; g.kind = wk; g.stride_kind = N;
        if (N > 64) HIPCHK(ctx, qpn_launch_avi_solve_big(g, wbig, s));
        else HIPCHK(ctx, qpn_launch_avi_solve(g, s));
    }
    if (x_dev && !mfma_shape && !x_in_kernel) {     // general sizes: strided device copy of the primal blocks (and to the replicas)
        HIPCHK(ctx, hipMemcpy2DAsync(x_dev, (size_t)stride_x * 8, d.z, (size_t)N * 8, (size_t)n * 8, (size_t)batch,
                                     hipMemcpyDeviceToDevice, s));
        for (int k = 0; mirrored && k < ctx->mirror_count; ++k)
            HIPCHK(ctx, hipMemcpy2DAsync(ctx->mirror_peer[k] + x_off, (size_t)stride_x * 8, d.z, (size_t)N * 8,
                                         (size_t)n * 8, (size_t)batch, hipMemcpyDefault, s));
    }
    return QPN_OK;
}

// One sweep: stages host buffers (the records only when h == null), carves the workspace, launches, reads back.
int solve_nodes_any(qpn_ctx *ctx, qpn_nodes *h, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                    const double *R, const double *qd, const double *Ad, const double *B, const double *l, const double *u,
                    const double *w, int64_t stride_w, double *z, int32_t *status, double *resid, int32_t *pivots,
                    uint8_t *active, const qpn_avi_opts *opts, int mem, double *x, int64_t stride_x)
{
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_solve_nodes: bad mem kind");
    const int N = n + m;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    qpn_avi_opts o;
    if (opts) o = *opts; else qpn_avi_default_opts(&o);
    hipStream_t s = ctx->stream;
    const size_t bN = (size_t)batch * N;
    const NodeSizes sz = node_sizes(batch, n, m, p, stride_w);
    const bool host = mem == QPN_MEM_HOST;
    const bool mfma_shape = n <= 32 && m <= 32 && m >= 1;
    // the handle may already know that the general path has nothing to do: no workspace for it then
    nodes_poll_declines(h);
    nodes_sync_route(ctx, h);
    // mid-size nodes: the fused kernels (no workspace); QPN_OPT_MID_ROUTE = 0 sends them down the general route
    const bool mid_ok = qpn_schur_wg_shape(n, m) && (o.max_pivots <= 0 || o.max_pivots - n >= 1);
    const bool wg2_ok = qpn_schur_wg2_shape(n, m) && (o.max_pivots <= 0 || o.max_pivots - n >= 1);
    const bool wg_shape = ctx->mid_route == 1 && (mid_ok || wg2_ok);
    const bool need_ws = !(h && (mfma_shape || wg_shape) && h->decl_state == 2);

    NodeDev d{Qd, R, qd, Ad, B, l, u, w, z, status, resid, pivots, active};
    double *wM = nullptr, *wq = nullptr, *wl = nullptr, *wu = nullptr, *wbig = nullptr; uint8_t *wk = nullptr;
    double *hQ, *hR, *hq, *hA, *hB, *hl, *hu, *hw, *hz = nullptr, *hres, *hx = nullptr; int32_t *hst, *hpv; uint8_t *hact;
    Carver cv(ctx);
    if (need_ws) {
        cv.add((void **)&wM, bN * N * 8); cv.add((void **)&wq, bN * 8); cv.add((void **)&wl, bN * 8);
        cv.add((void **)&wu, bN * 8); cv.add((void **)&wk, bN);
        if (N > 64) cv.add((void **)&wbig, qpn_avi_big_workspace_bytes(batch, N));
    }
    const bool z_ws = !z;          // no z wanted (handle calls): the kernels still need somewhere to put it
    if (host) {
        if (!h) {
            cv.add((void **)&hQ, sz.Q); cv.add((void **)&hR, sz.R + 8); cv.add((void **)&hq, sz.q);
            cv.add((void **)&hA, sz.A + 8); cv.add((void **)&hB, sz.B + 8); cv.add((void **)&hl, sz.lu + 8);
            cv.add((void **)&hu, sz.lu + 8);
        }
        cv.add((void **)&hw, sz.w + 8); cv.add((void **)&hz, bN * 8);
        cv.add((void **)&hres, (size_t)batch * 8); cv.add((void **)&hst, (size_t)batch * 4);
        cv.add((void **)&hpv, (size_t)batch * 4); cv.add((void **)&hact, bN);
        if (x && mfma_shape) cv.add((void **)&hx, (size_t)batch * n * 8);
    } else if (z_ws) cv.add((void **)&hz, bN * 8);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    if (host) {
        if (!h) {
            HIPCHK(ctx, hipMemcpyAsync(hQ, Qd, sz.Q, hipMemcpyHostToDevice, s));
            if (sz.R) HIPCHK(ctx, hipMemcpyAsync(hR, R, sz.R, hipMemcpyHostToDevice, s));
            HIPCHK(ctx, hipMemcpyAsync(hq, qd, sz.q, hipMemcpyHostToDevice, s));
            if (sz.A) HIPCHK(ctx, hipMemcpyAsync(hA, Ad, sz.A, hipMemcpyHostToDevice, s));
            if (sz.B) HIPCHK(ctx, hipMemcpyAsync(hB, B, sz.B, hipMemcpyHostToDevice, s));
            if (sz.lu) {
                HIPCHK(ctx, hipMemcpyAsync(hl, l, sz.lu, hipMemcpyHostToDevice, s));
                HIPCHK(ctx, hipMemcpyAsync(hu, u, sz.lu, hipMemcpyHostToDevice, s));
            }
            d.Q = hQ; d.R = hR; d.q = hq; d.A = hA; d.B = hB; d.l = hl; d.u = hu;
        }
        if (p > 0) HIPCHK(ctx, hipMemcpyAsync(hw, w, sz.w, hipMemcpyHostToDevice, s));
        if (z && !(o.flags & QPN_AVI_FLAG_COLD_START)) HIPCHK(ctx, hipMemcpyAsync(hz, z, bN * 8, hipMemcpyHostToDevice, s));
        else if (!z) o.flags |= QPN_AVI_FLAG_COLD_START;
        d.w = hw; d.z = hz; d.res = hres; d.st = hst; d.pv = hpv; d.act = hact;
    } else if (z_ws) { d.z = hz; o.flags |= QPN_AVI_FLAG_COLD_START; }

    double *x_dev = host ? hx : x;
    const int64_t sx_dev = host ? (int64_t)n : stride_x;
    rc = solve_nodes_launch(ctx, h, batch, n, m, p, d, stride_w, o, x_dev, sx_dev, wM, wq, wl, wu, wk, wbig, wg_shape);
    if (rc != QPN_OK) return rc;
    if (host) {
        if (z) HIPCHK(ctx, hipMemcpyAsync(z, d.z, bN * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(status, d.st, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
        if (resid) HIPCHK(ctx, hipMemcpyAsync(resid, d.res, (size_t)batch * 8, hipMemcpyDeviceToHost, s));
        if (pivots) HIPCHK(ctx, hipMemcpyAsync(pivots, d.pv, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
        if (active) HIPCHK(ctx, hipMemcpyAsync(active, d.act, bN, hipMemcpyDeviceToHost, s));
        if (x && hx)       // the primal blocks come down on their own (2.5 MB at 10 000 nodes; z is twice that)
            HIPCHK(ctx, hipMemcpy2DAsync(x, (size_t)stride_x * 8, hx, (size_t)n * 8, (size_t)n * 8, (size_t)batch,
                                         hipMemcpyDeviceToHost, s));
        else if (x)
            HIPCHK(ctx, hipMemcpy2DAsync(x, (size_t)stride_x * 8, d.z, (size_t)N * 8, (size_t)n * 8, (size_t)batch,
                                         hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
    }
    return QPN_OK;
}

int verify_nodes_any(qpn_ctx *ctx, bool records_on_device, int32_t batch, int32_t n, int32_t m, int32_t p,
                     const double *Qd, const double *R, const double *qd, const double *Ad, const double *B,
                     const double *l, const double *u, const double *xd, const double *w, int64_t stride_w, double tol,
                     int32_t *solution, double *lambda, int32_t *path, int mem);

} // namespace

extern "C" {

int qpn_solve_nodes_into(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                         const double *R, const double *qd, const double *Ad, const double *B,
                         const double *l, const double *u, const double *w, int64_t stride_w, double *z,
                         int32_t *status, double *resid, int32_t *pivots, uint8_t *active,
                         const qpn_avi_opts *opts, int mem, double *x, int64_t stride_x)
{
    if (!ctx) return QPN_ERR_ARG;
    if (x && stride_x < n) return fail_arg(ctx, "qpn_solve_nodes_into: stride_x < n");
    if (batch < 0 || n <= 0 || m < 0 || p < 0) return fail_arg(ctx, "qpn_solve_nodes: bad sizes");
    if (batch == 0) return QPN_OK;
    if (!Qd || !qd || (m > 0 && (!Ad || !l || !u)) || (p > 0 && (!R || !w || (m > 0 && !B))) || !z || !status)
        return fail_arg(ctx, "qpn_solve_nodes: null pointer");
    if (stride_w != 0 && stride_w < p) return fail_arg(ctx, "qpn_solve_nodes: stride_w < p");
    if (n + m > qpn_avi_max_n()) { ctx->last_error = "qpn_solve_nodes: n+m > 1024 not supported by ABI v1"; return QPN_ERR_SIZE; }
    return solve_nodes_any(ctx, nullptr, batch, n, m, p, Qd, R, qd, Ad, B, l, u, w, stride_w, z, status, resid, pivots,
                           active, opts, mem, x, stride_x);
}

// ---- resident node records ------------------------------------------------------------------------------------
int qpn_nodes_free(qpn_ctx *ctx, qpn_nodes *h)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!h) return QPN_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (h->decl_ev) { (void)hipEventSynchronize(h->decl_ev); (void)hipEventDestroy(h->decl_ev); }
    if (h->buf) (void)hipFree(h->buf);
    if (h->decl_dev) (void)hipFree(h->decl_dev);
    if (h->decl_host) (void)hipHostFree(h->decl_host);
    if (h->order) (void)hipFree(h->order);
    if (h->key) (void)hipFree(h->key);
    delete h;
    return QPN_OK;
}

int qpn_nodes_upload(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                     const double *R, const double *qd, const double *Ad, const double *B, const double *l,
                     const double *u, int mem, qpn_nodes **out)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!out) return fail_arg(ctx, "qpn_nodes_upload: null out");
    *out = nullptr;
    if (batch <= 0 || n <= 0 || m < 0 || p < 0) return fail_arg(ctx, "qpn_nodes_upload: bad sizes");
    if (!Qd || !qd || (m > 0 && (!Ad || !l || !u)) || (p > 0 && (!R || (m > 0 && !B))))
        return fail_arg(ctx, "qpn_nodes_upload: null pointer");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_nodes_upload: bad mem kind");
    if (n + m > qpn_avi_max_n()) { ctx->last_error = "qpn_nodes_upload: n+m > 1024 not supported by ABI v1"; return QPN_ERR_SIZE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    qpn_nodes *h = new (std::nothrow) qpn_nodes();
    if (!h) return fail_arg(ctx, "qpn_nodes_upload: out of host memory");
    h->device = ctx->device; h->batch = batch; h->n = n; h->m = m; h->p = p;
    const NodeSizes sz = node_sizes(batch, n, m, p, 0);
    const size_t fb[7] = {sz.Q, sz.R, sz.q, sz.A, sz.B, sz.lu, sz.lu};
    const double *src[7] = {Qd, R, qd, Ad, B, l, u};
    size_t total = 0, off[7];
    for (int i = 0; i < 7; ++i) { off[i] = total; total += (fb[i] + 8 + 255) & ~(size_t)255; h->fbytes[i] = fb[i]; }
    hipError_t e = hipMalloc((void **)&h->buf, total);
    if (e == hipSuccess) e = hipMalloc((void **)&h->decl_dev, 4);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->decl_host, 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->decl_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&h->order, (size_t)batch * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&h->key, (size_t)batch * 4);
    if (e == hipSuccess) e = hipMemsetAsync(h->key, 0, (size_t)batch * 4, ctx->stream);
    const hipMemcpyKind kind = mem == QPN_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    for (int i = 0; i < 7 && e == hipSuccess; ++i) {
        h->f[i] = reinterpret_cast<double *>(reinterpret_cast<char *>(h->buf) + off[i]);
        if (fb[i]) e = hipMemcpyAsync(h->f[i], src[i], fb[i], kind, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // the caller's arrays are free again on return
    if (e == hipSuccess) e = nodes_check_symmetry(ctx, h);
    if (e != hipSuccess) { int rc = fail_hip(ctx, e, "qpn_nodes_upload"); qpn_nodes_free(ctx, h); return rc; }
    *h->decl_host = 0;
    *out = h;
    return QPN_OK;
}

int qpn_nodes_update(qpn_ctx *ctx, qpn_nodes *h, int32_t field, const double *data, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!h || field < 0 || field > 6 || !data) return fail_arg(ctx, "qpn_nodes_update: bad argument");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_nodes_update: bad mem kind");
    if (h->device != ctx->device) return fail_arg(ctx, "qpn_nodes_update: handle belongs to another device");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (h->fbytes[field])
        HIPCHK(ctx, hipMemcpyAsync(h->f[field], data, h->fbytes[field],
                                   mem == QPN_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    if (mem == QPN_MEM_HOST) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // what was known about the old records is void (an answer still in flight must not be read as the new one's)
    if (h->decl_state == 1) HIPCHK(ctx, hipEventSynchronize(h->decl_ev));
    h->decl_state = 0;
    if (field == QPN_NODE_QD) HIPCHK(ctx, nodes_check_symmetry(ctx, h));
    return QPN_OK;
}

int qpn_nodes_set_schedule(qpn_ctx *ctx, qpn_nodes *h, int32_t period)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!h || period < 0) return fail_arg(ctx, "qpn_nodes_set_schedule: bad argument");
    h->period = period; h->calls = 0;
    if (period == 0) h->order_valid = false;
    return QPN_OK;
}

int qpn_nodes_info(qpn_ctx *ctx, qpn_nodes *h, int32_t info[4])
{
    if (!ctx) return QPN_ERR_ARG;
    if (!h || !info) return fail_arg(ctx, "qpn_nodes_info: null argument");
    nodes_poll_declines(h);
    info[0] = h->decl_state; info[1] = h->decl_state >= 2 ? *h->decl_host : 0;
    info[2] = (h->order_valid ? 1 : 0) | (h->sym ? 2 : 0); info[3] = h->calls;
    return QPN_OK;
}

int qpn_solve_nodes_h(qpn_ctx *ctx, qpn_nodes *h, const double *w, int64_t stride_w, double *z,
                      int32_t *status, double *resid, int32_t *pivots, uint8_t *active,
                      const qpn_avi_opts *opts, int mem, double *x, int64_t stride_x)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!h) return fail_arg(ctx, "qpn_solve_nodes_h: null handle");
    if (h->device != ctx->device) return fail_arg(ctx, "qpn_solve_nodes_h: handle belongs to another device");
    if (x && stride_x < h->n) return fail_arg(ctx, "qpn_solve_nodes_h: stride_x < n");
    if (!status || (h->p > 0 && !w)) return fail_arg(ctx, "qpn_solve_nodes_h: null pointer");
    if (stride_w != 0 && stride_w < h->p) return fail_arg(ctx, "qpn_solve_nodes_h: stride_w < p");
    return solve_nodes_any(ctx, h, h->batch, h->n, h->m, h->p, h->f[0], h->f[1], h->f[2], h->f[3], h->f[4], h->f[5], h->f[6],
                           w, stride_w, z, status, resid, pivots, active, opts, mem, x, stride_x);
}

int qpn_verify_nodes_h(qpn_ctx *ctx, qpn_nodes *h, const double *xd, const double *w, int64_t stride_w,
                       double tol, int32_t *solution, double *lambda, int32_t *path, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!h) return fail_arg(ctx, "qpn_verify_nodes_h: null handle");
    if (h->device != ctx->device) return fail_arg(ctx, "qpn_verify_nodes_h: handle belongs to another device");
    return verify_nodes_any(ctx, true, h->batch, h->n, h->m, h->p, h->f[0], h->f[1], h->f[2], h->f[3], h->f[4], h->f[5],
                            h->f[6], xd, w, stride_w, tol, solution, lambda, path, mem);
}

// ---- multi-GPU: shared iterate buffers and their replicas ------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) == QPN_IPC_HANDLE_BYTES, "IPC handle size");

int qpn_shared_alloc(qpn_ctx *ctx, size_t bytes, int flags, void **dev_ptr, uint8_t *handle)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!dev_ptr || !handle || bytes == 0) return fail_arg(ctx, "qpn_shared_alloc: null pointer or zero size");
    if (flags & ~QPN_SHARED_FINE_GRAINED) return fail_arg(ctx, "qpn_shared_alloc: unknown flag");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    if (flags & QPN_SHARED_FINE_GRAINED) HIPCHK(ctx, hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    else HIPCHK(ctx, hipMalloc(&p, bytes));
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return fail_hip(ctx, e, "hipIpcGetMemHandle"); }
    e = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(p); return fail_hip(ctx, e, "qpn_shared_alloc: clear"); }
    memcpy(handle, &h, sizeof h);
    *dev_ptr = p;
    return QPN_OK;
}

int qpn_shared_open(qpn_ctx *ctx, const uint8_t *handle, void **dev_ptr)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!dev_ptr || !handle) return fail_arg(ctx, "qpn_shared_open: null pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    void *p = nullptr;
    HIPCHK(ctx, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *dev_ptr = p;
    return QPN_OK;
}

int qpn_shared_close(qpn_ctx *ctx, void *dev_ptr)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!dev_ptr) return QPN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < ctx->mirror_count; ++k)
        if ((void *)ctx->mirror_peer[k] == dev_ptr) { ctx->mirror_count = 0; break; }   // never leave a dangling replica
    HIPCHK(ctx, hipIpcCloseMemHandle(dev_ptr));
    return QPN_OK;
}

int qpn_shared_free(qpn_ctx *ctx, void *dev_ptr)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!dev_ptr) return QPN_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((const void *)ctx->mirror_own == dev_ptr) { ctx->mirror_count = 0; ctx->mirror_own = nullptr; ctx->mirror_bytes = 0; }
    HIPCHK(ctx, hipFree(dev_ptr));
    return QPN_OK;
}

int qpn_set_primal_mirrors(qpn_ctx *ctx, const double *own, size_t bytes, int32_t count, double *const *peers)
{
    if (!ctx) return QPN_ERR_ARG;
    if (count < 0 || count > QPN_MAX_MIRRORS) return fail_arg(ctx, "qpn_set_primal_mirrors: count outside 0..QPN_MAX_MIRRORS");
    if (count > 0 && (!own || !peers || bytes == 0)) return fail_arg(ctx, "qpn_set_primal_mirrors: null pointer");
    for (int k = 0; k < count; ++k)
        if (!peers[k] || peers[k] == own) return fail_arg(ctx, "qpn_set_primal_mirrors: null or self peer");
    ctx->mirror_count = 0;
    ctx->mirror_own = count ? own : nullptr;
    ctx->mirror_bytes = count ? bytes : 0;
    for (int k = 0; k < count; ++k) ctx->mirror_peer[k] = peers[k];
    ctx->mirror_count = count;
    return QPN_OK;
}

static_assert(QPN_SWEEP_BOX_BYTES == 2 * QPN_MAX_RANKS * 32, "mailbox = 2 parities x QPN_MAX_RANKS slots of 32 B");

int qpn_sweep_status(qpn_ctx *ctx, const int32_t *status, const double *resid, int32_t count, double *out,
                     int32_t rank, int32_t world, void *const *boxes, uint64_t epoch, int32_t timeout_ms)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!status || !out || count < 0) return fail_arg(ctx, "qpn_sweep_status: bad arguments");
    if (world > QPN_MAX_RANKS) return fail_arg(ctx, "qpn_sweep_status: world > QPN_MAX_MIRRORS + 1");
    SweepBoxes bx{};
    if (world > 1) {
        if (rank < 0 || rank >= world || !boxes || epoch == 0 || timeout_ms <= 0)
            return fail_arg(ctx, "qpn_sweep_status: bad rank / boxes / epoch / timeout");
        for (int r = 0; r < world; ++r) {
            if (!boxes[r]) return fail_arg(ctx, "qpn_sweep_status: null mailbox");
            bx.box[r] = boxes[r];
        }
    } else { world = 1; rank = 0; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, qpn_launch_sweep_status(status, resid, count, out, rank, world, bx, epoch,
                                        (unsigned long long)timeout_ms * 100000ull, ctx->stream));   // 100 MHz wall clock
    return QPN_OK;
}

namespace {
int order_reserve(qpn_ctx *ctx, int32_t count)
{
    if (count <= ctx->order_cap) return QPN_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));          // earlier launches may still read the old buffer
    if (ctx->order) { HIPCHK(ctx, hipFree(ctx->order)); ctx->order = nullptr; ctx->order_cap = 0; ctx->order_count = 0; }
    HIPCHK(ctx, hipMalloc((void **)&ctx->order, (size_t)count * 4));
    ctx->order_cap = count;
    return QPN_OK;
}
} // namespace

int qpn_order_nodes_by_pivots(qpn_ctx *ctx, const int32_t *pivots, int32_t count, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!pivots || count <= 0) return fail_arg(ctx, "qpn_order_nodes_by_pivots: bad arguments");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_order_nodes_by_pivots: bad mem kind");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = order_reserve(ctx, count);
    if (rc != QPN_OK) return rc;
    const int32_t *dp = pivots;
    if (mem == QPN_MEM_HOST) {
        int32_t *tmp;
        Carver cv(ctx);
        cv.add((void **)&tmp, (size_t)count * 4);
        rc = cv.commit();
        if (rc != QPN_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(tmp, pivots, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
        dp = tmp;
    }
    HIPCHK(ctx, qpn_launch_order_by_pivots(dp, count, ctx->order, ctx->stream));
    if (mem == QPN_MEM_HOST) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->order_count = count;
    ctx->order_user = true;
    return QPN_OK;
}

int qpn_set_node_order(qpn_ctx *ctx, const int32_t *order, int32_t count, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (!order) { ctx->order_count = 0; ctx->order_user = false; ctx->auto_calls = 0; return QPN_OK; }
    if (count <= 0) return fail_arg(ctx, "qpn_set_node_order: bad count");
    if (mem != QPN_MEM_HOST && mem != QPN_MEM_DEVICE) return fail_arg(ctx, "qpn_set_node_order: bad mem kind");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = order_reserve(ctx, count);
    if (rc != QPN_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->order, order, (size_t)count * 4,
                               mem == QPN_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    if (mem == QPN_MEM_HOST) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->order_count = count;
    ctx->order_user = true;
    return QPN_OK;
}

int qpn_ctx_set_auto_schedule(qpn_ctx *ctx, int32_t period)
{
    if (!ctx) return QPN_ERR_ARG;
    if (period < 0) return fail_arg(ctx, "qpn_ctx_set_auto_schedule: negative period");
    ctx->auto_period = period;
    ctx->auto_calls = 0;
    if (period == 0 && !ctx->order_user) ctx->order_count = 0;      // drop a hint this mechanism installed
    return QPN_OK;
}

int qpn_verify_nodes(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p,
                     const double *Qd, const double *R, const double *qd, const double *Ad,
                     const double *B, const double *l, const double *u, const double *xd,
                     const double *w, int64_t stride_w, double tol, int32_t *solution,
                     double *lambda, int32_t *path, int mem)
{
    return verify_nodes_any(ctx, mem == QPN_MEM_DEVICE, batch, n, m, p, Qd, R, qd, Ad, B, l, u, xd, w, stride_w, tol, solution,
                            lambda, path, mem);
}

} // extern "C"

namespace {
int verify_nodes_any(qpn_ctx *ctx, bool records_on_device, int32_t batch, int32_t n, int32_t m, int32_t p,
                     const double *Qd, const double *R, const double *qd, const double *Ad, const double *B,
                     const double *l, const double *u, const double *xd, const double *w, int64_t stride_w, double tol,
                     int32_t *solution, double *lambda, int32_t *path, int mem)
{
    if (!ctx) return QPN_ERR_ARG;
    if (batch < 0 || n <= 0 || m < 0 || p < 0) return fail_arg(ctx, "qpn_verify_nodes: bad sizes");
    if (batch == 0) return QPN_OK;
    if (n > qpn_verify_max_dim() || m > qpn_verify_max_dim()) { ctx->last_error = "qpn_verify_nodes: n, m <= 512 in ABI v1"; return QPN_ERR_SIZE; }
    const bool wide_avi = m > 64;         // the bounded-LSQ fallback of wide nodes runs on the large-item AVI kernel
    double *wbig = nullptr;
    if (!Qd || !qd || !xd || (m > 0 && (!Ad || !l || !u || !lambda)) || (p > 0 && (!R || !w || (m > 0 && !B))) ||
        !solution || !path)
        return fail_arg(ctx, "qpn_verify_nodes: null pointer");
    if (stride_w != 0 && stride_w < p) return fail_arg(ctx, "qpn_verify_nodes: stride_w < p");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t mm = (size_t)(m > 0 ? m : 1);
    // scratch of the bounded-LSQ fallback (src/qp_processing.jl:129-137): Gram block + vectors
    double *sG, *sq, *slb, *sub, *sz, *sres, *gws = nullptr; int32_t *sst;
    const bool wide = n > QPN_VERIFY_WIDE_FROM || m > QPN_VERIFY_WIDE_FROM;
    const bool mid = !wide && (n > 32 || m > 32);      // verify_node64's class: a small slot workspace for the nodes it hands on
    const size_t mp16 = (size_t)((m + 15) & ~15);
    if (mem == QPN_MEM_DEVICE) {
        Carver cv(ctx);
        cv.add((void **)&sG, (size_t)batch * mm * mm * 8); cv.add((void **)&sq, (size_t)batch * mm * 8);
        cv.add((void **)&slb, (size_t)batch * mm * 8); cv.add((void **)&sub, (size_t)batch * mm * 8);
        cv.add((void **)&sz, (size_t)batch * mm * 8); cv.add((void **)&sres, (size_t)batch * 8);
        cv.add((void **)&sst, (size_t)batch * 4);
        if (wide_avi) cv.add((void **)&wbig, qpn_avi_big_workspace_bytes(batch, m));
        if (wide && m > 0) cv.add((void **)&gws, (size_t)batch * 2 * mp16 * mp16 * 8);
        else if (mid && m > 0) cv.add((void **)&gws, (size_t)QPN_VERIFY_MID_SLOTS * 2 * mp16 * mp16 * 8 + 64);
        int rc = cv.commit();
        if (rc != QPN_OK) return rc;
        HIPCHK(ctx, qpn_launch_verify_nodes(batch, n, m, p, Qd, R, qd, Ad, B, l, u, xd, w, stride_w, tol,
                                            solution, lambda, path, sG, sq, slb, sub, sz, sres, sst, s, wbig, gws));
        return QPN_OK;
    }
    if (mem != QPN_MEM_HOST) return fail_arg(ctx, "qpn_verify_nodes: bad mem kind");
    const NodeSizes sz_ = node_sizes(batch, n, m, p, stride_w);
    const double *dQ = Qd, *dR = R, *dq = qd, *dA = Ad, *dB = B, *dl = l, *du = u;
    double *sQ, *sR, *sqq, *sA, *sB, *sl, *su;
    double *dw, *dx, *dlam; int32_t *dsol, *dpath;
    Carver cv(ctx);
    if (!records_on_device) {
        cv.add((void **)&sQ, sz_.Q); cv.add((void **)&sR, sz_.R + 8); cv.add((void **)&sqq, sz_.q);
        cv.add((void **)&sA, sz_.A + 8); cv.add((void **)&sB, sz_.B + 8); cv.add((void **)&sl, sz_.lu + 8);
        cv.add((void **)&su, sz_.lu + 8);
    }
    cv.add((void **)&dw, sz_.w + 8); cv.add((void **)&dx, sz_.q);
    cv.add((void **)&dlam, sz_.lu + 8); cv.add((void **)&dsol, (size_t)batch * 4);
    cv.add((void **)&dpath, (size_t)batch * 4);
    cv.add((void **)&sG, (size_t)batch * mm * mm * 8); cv.add((void **)&sq, (size_t)batch * mm * 8);
    cv.add((void **)&slb, (size_t)batch * mm * 8); cv.add((void **)&sub, (size_t)batch * mm * 8);
    cv.add((void **)&sz, (size_t)batch * mm * 8); cv.add((void **)&sres, (size_t)batch * 8);
    cv.add((void **)&sst, (size_t)batch * 4);
    if (wide_avi) cv.add((void **)&wbig, qpn_avi_big_workspace_bytes(batch, m));
    if (wide && m > 0) cv.add((void **)&gws, (size_t)batch * 2 * mp16 * mp16 * 8);
    else if (mid && m > 0) cv.add((void **)&gws, (size_t)QPN_VERIFY_MID_SLOTS * 2 * mp16 * mp16 * 8 + 64);
    int rc = cv.commit();
    if (rc != QPN_OK) return rc;
    if (!records_on_device) {
        HIPCHK(ctx, hipMemcpyAsync(sQ, Qd, sz_.Q, hipMemcpyHostToDevice, s));
        if (sz_.R) HIPCHK(ctx, hipMemcpyAsync(sR, R, sz_.R, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(sqq, qd, sz_.q, hipMemcpyHostToDevice, s));
        if (sz_.A) HIPCHK(ctx, hipMemcpyAsync(sA, Ad, sz_.A, hipMemcpyHostToDevice, s));
        if (sz_.B) HIPCHK(ctx, hipMemcpyAsync(sB, B, sz_.B, hipMemcpyHostToDevice, s));
        if (sz_.lu) {
            HIPCHK(ctx, hipMemcpyAsync(sl, l, sz_.lu, hipMemcpyHostToDevice, s));
            HIPCHK(ctx, hipMemcpyAsync(su, u, sz_.lu, hipMemcpyHostToDevice, s));
        }
        dQ = sQ; dR = sR; dq = sqq; dA = sA; dB = sB; dl = sl; du = su;
    }
    HIPCHK(ctx, hipMemcpyAsync(dx, xd, sz_.q, hipMemcpyHostToDevice, s));
    if (p > 0) HIPCHK(ctx, hipMemcpyAsync(dw, w, sz_.w, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, qpn_launch_verify_nodes(batch, n, m, p, dQ, dR, dq, dA, dB, dl, du, dx, dw, stride_w, tol,
                                        dsol, dlam, dpath, sG, sq, slb, sub, sz, sres, sst, s, wbig, gws));
    HIPCHK(ctx, hipMemcpyAsync(solution, dsol, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(path, dpath, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
    if (sz_.lu) HIPCHK(ctx, hipMemcpyAsync(lambda, dlam, sz_.lu, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return QPN_OK;
}

} // namespace
