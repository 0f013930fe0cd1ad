// Internal declarations shared by the HIP translation units of libqpn_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/qpn_hip.h"

// Developer switches on the environment (kernel A/B, diagnostic routes) exist in diagnostic builds only
// (csrc/build.sh -DQPN_DEV_SWITCHES): the product library dispatches on its arguments and on qpn_ctx_set_option alone and
// never reads the environment.
#ifdef QPN_DEV_SWITCHES
#include <stdlib.h>
#define QPN_DEV_ENV(name) getenv(name)
#else
#define QPN_DEV_ENV(name) (static_cast<const char *>(nullptr))
#endif

// per-node records (qpn_solve_nodes): the solve kernel assembles the KKT blocks on the fly
struct NodeSrc {
    int32_t n, m, p;
    const double *Qd, *R, *qd, *Ad, *B, *l, *u, *w;
    int64_t stride_w;
    int32_t sym;     // 1 = every Qd of these records is bitwise symmetric (known for resident records only)
};

// internal flag bits of AviBatchArgs::flags (above the ABI's QPN_AVI_FLAG_*)
// register kernel only: M is given in the register-block layout of avi_solve_reg<BS> -- lane fastest, [l][k][lane] with BS x BS
// entries per lane, item stride strideM --, every row is a GAVI row, cold start; the kernel then loads its dictionary with
// plain coalesced loads (no LDS staging), and returns z / status / pivots without the post-check (the caller checks on its
// own original blocks)
#define QPN_AVI_IFLAG_BLOCKED_M (1 << 16)
int qpn_avi_reg_block_size(int N);      // BS of the instantiation qpn_launch_avi_solve_reg picks for size N

struct AviBatchArgs {
    int32_t batch;
    int32_t N;
    const double *M;
    int64_t strideM;
    const double *q;
    const double *l;
    const double *u;
    const uint8_t *kind;
    int64_t stride_kind;
    double *z;
    int32_t *status;
    double *resid;
    int32_t *pivots;
    uint8_t *active;
    double check_tol, piv_tol, feas_tol, comp_tol;
    int32_t max_pivots;
    int32_t flags;   // QPN_AVI_FLAG_*
    // optional gate: item b runs only when only_if[b] == only_if_value (others are left untouched)
    const int32_t *only_if;
    int32_t only_if_value;
    // compact form of the gate (register kernel only): scan != 0 launches a SMALL grid whose waves scan
    // only_if[] and solve just the matching items, one after the other; with assemble_first != 0 the
    // item's blocks are first assembled from `nd` into M/q/l/u/kind (the fallback of qpn_solve_nodes).
    int32_t scan;
    int32_t assemble_first;
    // diagnostic builds only (-DQPN_STAMPS): per-item cycle sums per phase, [batch][8] uint64
    unsigned long long *stamps;
    NodeSrc nd;      // used by the fused node path only
    // optional scatter of the primal block z[b][0..nd.n) into x[b * stride_x + i] (qpn_solve_nodes_into)
    double *x;
    int64_t stride_x;
    // replicas of the iterate on peer GPUs (qpn_set_primal_mirrors): every primal block stored to x[off] is
    // also stored to mirror[k][off], k < n_mirror -- plain stores over xGMI, no collective
    int32_t n_mirror;
    double *mirror[QPN_MAX_MIRRORS];
    // optional schedule of the fused node kernel: wavefront i solves node order[i] (a permutation)
    const int32_t *order;
    // large-item kernel only: per-item size override (items of different N <= this->N in one launch; vectors
    // are then laid out with stride vec_stride, the workspace with the launch-wide N)
    const int32_t *n_items;
    int64_t vec_stride;
    // fused node kernel only: counts the items it declines (status = -1), for callers that own the records and
    // want to know whether the general-kernel launch behind it has anything to do (qpn_nodes handles); may be null
    int32_t *decl_count;
    // fused node kernel only: exponentially smoothed pivot count per node (units of 1/32 pivot), updated by every sweep
    // (key <- key - key/32 + pivots); the longest-first order of a resident-records handle is made from it.  May be null.
    int32_t *sched_key;
};

// Function attributes (dynamic LDS limits) are per device: a launcher sets them once per device it is used on.
// (Two host threads racing here both set the same value.)
struct QpnPerDeviceOnce {
    bool done[64] = {};
    int device() const { int d = 0; (void)hipGetDevice(&d); return d & 63; }
};

// qpn_avi_solve.hip
hipError_t qpn_launch_avi_solve(const AviBatchArgs &a, hipStream_t stream);      // dispatcher
hipError_t qpn_launch_avi_solve_lds1(const AviBatchArgs &a, hipStream_t stream); // LDS-tableau kernel
int qpn_avi_max_n();

// qpn_avi_big.hip: 64 < N <= 1024, one workgroup per item, dictionary in an HBM workspace
int qpn_avi_big_max_n();
size_t qpn_avi_big_workspace_bytes(int batch, int N);
hipError_t qpn_launch_avi_solve_big(const AviBatchArgs &a, double *workspace, hipStream_t stream);

// A8: nodes with n or m above this take the workgroup kernel verify_wide_node (qpn_verify.hip), the others one wavefront per node
#ifndef QPN_VERIFY_WIDE_FROM
#define QPN_VERIFY_WIDE_FROM 64
#endif
// ... and the few nodes of the 33 .. 64 class with more than 32 active rows (verify_node64 flags them) take the same kernel out
// of a workspace of this many slots (a device counter hands them out; nodes beyond it go to round 1's kernels)
constexpr int QPN_VERIFY_MID_SLOTS = 256;

// qpn_avi_schur_big.hip: blocked MFMA crash for large node-shaped items (workspace views filled by stage A)
struct SchurBigWs {
    double *Tt, *S, *c, *l2, *u2, *lam;
    int32_t *st2, *piv2, *nsplit, *nred;
    int64_t tt_stride, s_stride;
    int32_t s_rowmajor;      // layout of S (2: already in the Lemke kernel's dictionary, row stride m + 1): row-major for the workgroup Lemke kernel (its dictionary is row-major: no
                             // transposition anywhere), column-major for the kernels that take S as an ABI-layout M
};
size_t qpn_schur_big_workspace_bytes(int batch, int N);
hipError_t qpn_launch_schur_big_stage_a(const AviBatchArgs &a, void *ws, SchurBigWs *out, int s_rowmajor, hipStream_t stream);
hipError_t qpn_launch_schur_big_finish(const AviBatchArgs &a, const SchurBigWs &w, hipStream_t stream);
hipError_t qpn_launch_schur_big_lemke(const AviBatchArgs &a, const SchurBigWs &w, double *dict, hipStream_t stream, int after_bpp = 0);
hipError_t qpn_launch_schur_big_bpp(const AviBatchArgs &a, const SchurBigWs &w, double *dict, hipStream_t stream);   // symmetric S only

// qpn_avi_schur_big2.hip: the same stage A straight from node records (64 < n <= 256, m <= 256), rank-64 block pivots, one
// workgroup of 16 wavefronts per node; its finish reads the records too.  Same workspace views: the Lemke kernel above runs
// between the two.
bool qpn_schur_big2_shape(int n, int m);
hipError_t qpn_launch_schur_big2_stage_a(const AviBatchArgs &a, void *ws, double *dict, SchurBigWs *out, hipStream_t stream);   // a.nd set
hipError_t qpn_launch_schur_big2_finish(const AviBatchArgs &a, const SchurBigWs &w, hipStream_t stream);

// qpn_avi_schur_wg.hip: node records with 49 <= max(n, m) <= 64 in ONE fused workgroup kernel (crash, Lemke, read-back, post-check; no workspace)
bool qpn_schur_wg_shape(int n, int m);
hipError_t qpn_launch_schur_wg_nodes(const AviBatchArgs &a, hipStream_t stream);
// one wavefront per node, 33 <= max(n, m) <= 48 (qpn_avi_schur48.hip)
bool qpn_schur48_shape(int n, int m);
hipError_t qpn_launch_avi_solve_schur48_nodes(const AviBatchArgs &a, hipStream_t stream);              // a.nd set; declined nodes keep status -1

// qpn_avi_schur_wg2.hip: the fused workgroup kernel for n, m <= 128 (one of them > 64): two wavefronts per row tile
bool qpn_schur_wg2_shape(int n, int m);
hipError_t qpn_launch_schur_wg2_nodes(const AviBatchArgs &a, hipStream_t stream);             // a.nd set; declined nodes keep status -1

// qpn_avi_schur.hip: MFMA Schur-complement variant for items of shape [free STD x n | GAVI x m]
hipError_t qpn_launch_avi_solve_schur(const AviBatchArgs &a, double *dbgS, double *dbgc, double *dbgW,
                                      double *dbgh, hipStream_t stream);
hipError_t qpn_launch_avi_solve_schur_nodes(const AviBatchArgs &a, hipStream_t stream);   // a.nd set, a.M unused

// qpn_avi_reg.hip
hipError_t qpn_launch_avi_solve_reg(const AviBatchArgs &a, hipStream_t stream);  // register-tableau kernel

// qpn_kkt.hip
hipError_t qpn_launch_order_by_pivots(const int32_t *pivots, int32_t count, int32_t *order, hipStream_t stream,
                                      int32_t *key = nullptr);     // key: smoothed counts, see the kernel
// per-sweep status pair, optionally exchanged through the ranks' mailboxes (box[r] = rank r's mailbox as mapped here)
#define QPN_MAX_RANKS (QPN_MAX_MIRRORS + 1)
struct SweepBoxes { void *box[QPN_MAX_RANKS]; };
hipError_t qpn_launch_sweep_status(const int32_t *status, const double *resid, int32_t count, double *out,
                                   int32_t rank, int32_t world, const SweepBoxes &boxes, unsigned long long epoch,
                                   unsigned long long timeout_ticks, hipStream_t stream);
// local pieces (local_piece) and recipe enumeration (all_Ks): all pointers device
// flag[0] = 1 if any Qd block (n x n, column-major, `batch` of them) is not bitwise symmetric; flag[0] is left alone otherwise
hipError_t qpn_launch_qd_asymmetry(int32_t batch, int32_t n, const double *Qd, int32_t *flag, hipStream_t stream);
hipError_t qpn_launch_local_pieces(int32_t batch, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd, const double *R,
                                   const double *qd, const double *Ad, const double *B, const double *l, const double *u,
                                   const int32_t *node_of, const uint8_t *K, double *Ap, double *lp, double *up, uint8_t *keep,
                                   hipStream_t stream);
hipError_t qpn_launch_recipes(int32_t N, const uint8_t *mask, long long first, int32_t count, uint8_t *K, hipStream_t stream);
// qpn_pieces.hip: a level's recipes / pieces with the multipliers eliminated, one call each
hipError_t qpn_launch_recipes_batch(int32_t nodes, int32_t N, const uint8_t *masks, const long long *offsets, long long total, uint8_t *K,
                                    int32_t *node_of, hipStream_t stream);
size_t qpn_reduce_pieces_lds(int32_t n, int32_t m, int32_t p);
hipError_t qpn_launch_reduce_pieces(int32_t pieces, int32_t n, int32_t m, int32_t p, double tol, double *Ap, const double *lp,
                                    const double *up, const uint8_t *keep, double *Ar, double *lr, double *ur, int32_t *rows_out,
                                    int32_t *flags_out, hipStream_t stream);
// pool assembly (combine_gavis): all pointers device
struct QpnPoolLaunch {
    int32_t batch, form, nd, sn, sm, p;
    const int32_t *xi_owner, *xi_dpos, *con_owner, *dec_src;
    const double *Qd, *Qp, *qd, *Ad, *Bp, *l, *u, *w;
    int64_t s_Qd, s_Qp, s_qd, s_Ad, s_Bp, s_lu, s_w;
    double *M, *q, *lo, *hi; uint8_t *kind;
    int64_t s_M;
};
hipError_t qpn_launch_assemble_pools(const QpnPoolLaunch &L, hipStream_t stream);
hipError_t qpn_launch_check_avi(int32_t batch, int32_t N, const double *M, int64_t strideM,
                                const double *q, const double *l, const double *u,
                                const uint8_t *kind, int64_t stride_kind, const double *z,
                                double tol, int32_t *degree, double *r, hipStream_t stream);
hipError_t qpn_launch_comp_indices(int64_t count, const double *zv, const double *rv,
                                   const double *l, const double *u, double tol, int32_t shift,
                                   uint8_t *mask, hipStream_t stream);
hipError_t qpn_launch_assemble_nodes(int32_t batch, int32_t n, int32_t m, int32_t p,
                                     const double *Qd, const double *R, const double *qd,
                                     const double *Ad, const double *B, const double *l,
                                     const double *u, const double *w, int64_t stride_w,
                                     double *Mout, double *qout, double *lout, double *uout,
                                     uint8_t *kind_out, hipStream_t stream,
                                     const int32_t *only_if = nullptr, int32_t only_if_value = 0);
hipError_t qpn_launch_verify_nodes(int32_t batch, int32_t n, int32_t m, int32_t p,
                                   const double *Qd, const double *R, const double *qd,
                                   const double *Ad, const double *B, const double *l,
                                   const double *u, const double *xd, const double *w,
                                   int64_t stride_w, double tol, int32_t *solution, double *lambda,
                                   int32_t *path, double *sG, double *sq, double *slb, double *sub,
                                   double *sz, double *sres, int32_t *sst, hipStream_t stream,
                                   double *wbig = nullptr,       // wbig: large-item AVI workspace (m > 64 only)
                                   double *gws = nullptr);       // gws: [batch][2][pad16(m)^2] Gram block / factor of verify_wide_node (n or m > 64)
int qpn_verify_max_dim();

// ---- wave64 helpers (CDNA4: one wavefront = 64 lanes) --------------------------------
#ifdef __HIPCC__
// (A5+A6) reduced single-node KKT assembly of item b by ONE wavefront (src/avi.jl:205-251 + :305-377):
// M = [[Qd, -Ad'],[Ad, 0]] column-major, q = [qd + R w; B w], bounds [free; l..u], kinds [STD; GAVI].
// Column j of M (length N = n+m) is written by lanes striding the rows: coalesced.
__device__ __forceinline__ void qpn_assemble_item(const NodeSrc &nd, int b, int lane, double *Mout,
                                                  double *qout, double *lout, double *uout,
                                                  uint8_t *kind_out)
{
    const int n = nd.n, m = nd.m, p = nd.p, N = n + m;
    const double *Q_ = nd.Qd + (size_t)b * n * n;
    const double *A_ = nd.Ad + (size_t)b * m * n;
    const double *R_ = nd.R + (size_t)b * n * p;
    const double *B_ = nd.B + (size_t)b * m * p;
    const double *w_ = nd.w + (size_t)b * (size_t)nd.stride_w;
    double *Mo = Mout + (size_t)b * N * N;
    const size_t vo = (size_t)b * (size_t)N;
    // columns 0..n-1: [Qd(:,j) ; Ad(:,j)]
    for (int j = 0; j < n; ++j)
        for (int i = lane; i < N; i += 64)
            Mo[(size_t)j * N + i] = i < n ? Q_[(size_t)j * n + i] : A_[(size_t)j * m + (i - n)];
    // columns n..N-1: [-Ad(c,:)' ; 0]
    for (int c = 0; c < m; ++c)
        for (int i = lane; i < N; i += 64)
            Mo[(size_t)(n + c) * N + i] = i < n ? -A_[(size_t)i * m + c] : 0.0;
    for (int i = lane; i < N; i += 64) {
        double s;
        if (i < n) {
            s = nd.qd[(size_t)b * n + i];
            for (int k = 0; k < p; ++k) s = fma(R_[(size_t)k * n + i], w_[k], s);
            lout[vo + i] = -__builtin_huge_val(); uout[vo + i] = __builtin_huge_val(); kind_out[vo + i] = QPN_ROW_STD;
        } else {
            const int r = i - n;
            s = 0.0;
            for (int k = 0; k < p; ++k) s = fma(B_[(size_t)k * m + r], w_[k], s);
            lout[vo + i] = nd.l[(size_t)b * m + r]; uout[vo + i] = nd.u[(size_t)b * m + r];
            kind_out[vo + i] = QPN_ROW_GAVI;
        }
        qout[vo + i] = s;
    }
}

__device__ __forceinline__ double udbl(double v);
// ---- DPP / readlane based wave64 primitives (no LDS round trips) -------------------------
// dpp_ctrl encodings (GFX9): quad_perm = 0x00..0xFF, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every source lane of the row patterns used here is valid, so `old` is irrelevant: mov_dpp
    // (old = undef) saves the two register copies update_dpp(old = v) costs
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// value of `v` in lane `l`; l must be wave-uniform (it is turned into an SGPR)
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    l = __builtin_amdgcn_readfirstlane(l);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int readlane_i32(int v, int l)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}
// v with lane `lane` (wave-uniform) replaced by the wave-uniform value `x`: v_writelane_b32, no
// compare/select.  (No clang builtin on this toolchain, hence asm; gfx9 allows one SGPR per VALU instruction, so the lane
// select goes through M0 (saved and restored: the compiler owns M0); the s_nop covers the wait states between the SALU write of M0 / a VALU-written
// data SGPR and their use, which the hazard recognizer cannot see through inline asm.)
__device__ __forceinline__ int writelane_i32(int v, int lane, int x)
{
    const int xs = __builtin_amdgcn_readfirstlane(x), ls = __builtin_amdgcn_readfirstlane(lane);
    int m0save;
    asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
        : "+v"(v), "=&s"(m0save) : "s"(xs), "s"(ls));
    return v;
}
__device__ __forceinline__ double writelane_f64(double v, int lane, double x)
{
    const int ls = __builtin_amdgcn_readfirstlane(lane);
    const int xl = __builtin_amdgcn_readfirstlane(__double2loint(x)), xh = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    int lo = __double2loint(v), hi = __double2hiint(v);
    int m0save;
    asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\ts_nop 3\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\t"
        "s_mov_b32 m0, %2"
        : "+v"(lo), "+v"(hi), "=&s"(m0save) : "s"(xl), "s"(xh), "s"(ls));
    return __hiloint2double(hi, lo);
}
#define QPN_ROW_REDUCE(v, OP)                                  \
    do {                                                       \
        double o__;                                            \
        o__ = dpp_f64<0xB1>(v); v = OP(v, o__); /* quad_perm [1,0,3,2] */ \
        o__ = dpp_f64<0x4E>(v); v = OP(v, o__); /* quad_perm [2,3,0,1] */ \
        o__ = dpp_f64<0x141>(v); v = OP(v, o__); /* row_half_mirror */    \
        o__ = dpp_f64<0x140>(v); v = OP(v, o__); /* row_mirror      */    \
    } while (0)
__device__ __forceinline__ double qpn_max2(double a, double b) { return fmax(a, b); }  // v_max_f64
__device__ __forceinline__ double qpn_min2(double a, double b) { return fmin(a, b); }  // v_min_f64
// NaN-free inputs assumed by callers (they feed -1 / +inf sentinels for inactive lanes)
__device__ __forceinline__ double wave_max_f64(double v)
{
    QPN_ROW_REDUCE(v, qpn_max2);
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return udbl(qpn_max2(qpn_max2(r0, r1), qpn_max2(r2, r3)));
}
__device__ __forceinline__ double wave_min_f64(double v)
{
    QPN_ROW_REDUCE(v, qpn_min2);
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return udbl(qpn_min2(qpn_min2(r0, r1), qpn_min2(r2, r3)));
}
// min over lanes 0..31 only (two DPP rows); lanes 32..63 are ignored
__device__ __forceinline__ double wave_min32_f64(double v)
{
    QPN_ROW_REDUCE(v, qpn_min2);
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    return qpn_min2(r0, r1);
}
// v_min_f64 without the canonicalising v_max the compiler puts in front of fmin() (callers feed no NaNs).
// Inline asm is invisible to the hazard recognizer: the s_nop covers a DPP read of the result.
__device__ __forceinline__ double min_f64_nc(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min over lanes 0..31, returned in EVERY lane 0..31 (lanes 32..63 get the min of their own half):
// four DPP row stages + one ds_swizzle (lane ^ 16, on the LDS crossbar) -- no v_readlane, no SGPR detour
__device__ __forceinline__ double wave_min32_all_f64(double v)
{
    v = min_f64_nc(v, dpp_f64<0xB1>(v));
    v = min_f64_nc(v, dpp_f64<0x4E>(v));
    v = min_f64_nc(v, dpp_f64<0x141>(v));
    v = min_f64_nc(v, dpp_f64<0x140>(v));
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
    return min_f64_nc(v, __hiloint2double(hi, lo));
}
// the same minimum with the row crossing through v_readlane (two SGPR pairs) instead of the LDS crossbar: a few
// more instructions, but ~100 cycles less latency -- for loops whose waves wait on dependent chains, not on issue
__device__ __forceinline__ double wave_min32_all_lowlat_f64(double v)
{
    v = min_f64_nc(v, dpp_f64<0xB1>(v));
    v = min_f64_nc(v, dpp_f64<0x4E>(v));
    v = min_f64_nc(v, dpp_f64<0x141>(v));
    v = min_f64_nc(v, dpp_f64<0x140>(v));
    return min_f64_nc(readlane_f64(v, 0), readlane_f64(v, 16));
}
// min over lanes 0..31 of v and the wave-uniform `lim`, returned wave-uniform (it lands in an SGPR pair): four DPP row
// stages, then row 0's minimum crosses into row 1 with row_bcast:15 (gfx9 DPP: lane 15 of a row to every lane of the next
// one) and lane 31 is read out -- 2 DPP moves + 2 v_readlane for the crossing instead of 4 v_readlane + 4 v_mov; `lim`
// rides in as one v_min with a scalar operand before the stages instead of a v_mov + v_min after them
__device__ __forceinline__ double wave_min32_with_limit_f64(double v, double lim)
{
    {
        const double ls = udbl(lim);
        double r;
        asm("v_min_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(v), "s"(ls));
        v = r;
    }
    v = min_f64_nc(v, dpp_f64<0xB1>(v));
    v = min_f64_nc(v, dpp_f64<0x4E>(v));
    v = min_f64_nc(v, dpp_f64<0x141>(v));
    v = min_f64_nc(v, dpp_f64<0x140>(v));
    {
        // rows 1 and 3 receive lane 15 of rows 0 and 2 (row_mask 0xa)
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_mov_dpp(lo, 0x142, 0xA, 0xF, false);      // (rows 0, 2 of the result are undefined: only lane 31 is read)
        hi = __builtin_amdgcn_mov_dpp(hi, 0x142, 0xA, 0xF, false);
        v = min_f64_nc(v, __hiloint2double(hi, lo));
    }
    return readlane_f64(v, 31);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// lane mask of a predicate.  HIP's __ballot(p) compares int(p) with 0: the compiler materialises the flag in a VGPR and
// compares it again (v_cndmask + v_cmp_ne per ballot); the builtin takes the i1 as it is -- the compare that produced
// the predicate IS the mask
__device__ __forceinline__ unsigned long long qpn_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
// lowest lane whose predicate holds, or -1 (wave-uniform result)
__device__ __forceinline__ int wave_first(bool pred)
{
    unsigned long long b = qpn_ballot(pred);
    return b ? (__ffsll((long long)b) - 1) : -1;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Tell the compiler a value is wave-uniform (moves it to SGPRs): branches on it become scalar.
__device__ __forceinline__ bool ubool(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ double udbl(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
#endif
