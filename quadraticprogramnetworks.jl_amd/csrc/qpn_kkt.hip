// qpn_kkt.hip -- the array kernels around the AVI solve, for gfx950 (CDNA4):
//   check_avi      src/avi.jl:148-156          (A3)
//   comp_indices   src/avi_solutions.jl:511-562 (A9)
//   assemble_nodes src/avi.jl:205-251 + :305-377, single-node pools, reduced form (A5+A6)
// All are HBM-bound streaming kernels: one wavefront per item, coalesced column accesses
// (lane i <-> row i of a column-major block), no LDS staging needed.
#include "qpn_internal.h"

#define QINF __builtin_huge_val()

namespace {

constexpr int WAVE = 64;

// ---- (A3) check_avi_solution: one wave per item, lanes stride the rows -------------------
__global__ __launch_bounds__(256) void check_avi_kernel(int32_t batch, int32_t N, const double *M,
                                                        int64_t strideM, const double *q,
                                                        const double *l, const double *u,
                                                        const uint8_t *kind, int64_t stride_kind,
                                                        const double *z, double tol,
                                                        int32_t *degree, double *r_out)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int b = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (b >= batch) return;
    const double *Mg = M + (size_t)b * (size_t)strideM;
    const size_t vo = (size_t)b * (size_t)N;
    int bad = 0;
    for (int i = lane; i < N; i += WAVE) {
        double r = q[vo + i];
        for (int j = 0; j < N; ++j) {
            double zj = z[vo + j];
            if (zj != 0.0) r = fma(Mg[(size_t)j * N + i], zj, r);
        }
        if (r_out) r_out[vo + i] = r;
        const int g = kind ? (int)kind[(size_t)b * (size_t)stride_kind + i] : 0;
        const double zi = z[vo + i];
        const double p = g ? r : zi, d = g ? zi : r;
        const double li = l[vo + i], ui = u[vo + i];
        if (d > tol && fabs(p - li) > tol) bad++;   // :152
        if (d < -tol && fabs(p - ui) > tol) bad++;  // :153
        if (p - li < -tol) bad++;                   // :154
        if (p - ui > tol) bad++;
        if (isnan(p) || isnan(d)) bad++;
    }
    bad = wave_sum_i32(bad);
    if (lane == 0) degree[b] = bad;
}

// ---- (A9) comp_indices core: flat, one thread per row ------------------------------------
__global__ __launch_bounds__(256) void comp_indices_kernel(int64_t count, const double *zv,
                                                           const double *rv, const double *l,
                                                           const double *u, double tol,
                                                           int32_t shift, uint8_t *mask)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        const double z = zv[i], r = rv[i], li = l[i], ui = u[i];
        auto approx = [&](double x, double y) {
            return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= tol);
        };
        unsigned m = 0;
        if (!approx(li, ui)) {                                                   // :512
            if (approx(z, li) && r >= -tol) m |= 1u;                             // :543
            if (li - tol <= z && z <= ui + tol && fabs(r) <= tol) m |= 2u;       // :546
            if (approx(z, ui) && r <= tol) m |= 4u;                              // :549
        } else m = 8u;                                                           // :552-558
        mask[i] = (uint8_t)(m << shift);
    }
}

// ---- (A5+A6) reduced single-node KKT assembly: one wave per node ---------------------------
// Output column j of M_i (length N = n+m) is written by lanes striding the rows: coalesced.
__global__ __launch_bounds__(256) void assemble_nodes_kernel(
    int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd, const double *R,
    const double *qd, const double *Ad, const double *B, const double *l, const double *u,
    const double *w, int64_t stride_w, double *Mout, double *qout, double *lout, double *uout,
    uint8_t *kind_out, const int32_t *only_if, int32_t only_if_value)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int b = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (b >= batch) return;
    if (only_if && only_if[b] != only_if_value) return;   // gated (fallback of the fused node path)
    const NodeSrc nd{n, m, p, Qd, R, qd, Ad, B, l, u, w, stride_w};
    qpn_assemble_item(nd, b, lane, Mout, qout, lout, uout, kind_out);
}

// Large nodes (N > 64): one workgroup per (node, strip of 32 columns of M) instead of one wave per node -- a node's
// block is megabytes.  Columns j < n are two contiguous copies ([Qd(:,j); Ad(:,j)]); columns n + c are -Ad(c,:)'
// over zeros: a transpose, done through a 32 x 33 LDS tile so that both the reads (along c) and the writes (along
// the rows of M) are contiguous.  Strip 0 also writes q, l, u, kind (same fma order as qpn_assemble_item).
__global__ __launch_bounds__(256) void assemble_nodes_wide_kernel(
    int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd, const double *R,
    const double *qd, const double *Ad, const double *B, const double *l, const double *u,
    const double *w, int64_t stride_w, double *Mout, double *qout, double *lout, double *uout,
    uint8_t *kind_out, const int32_t *only_if, int32_t only_if_value)
{
    const int b = blockIdx.x, strip = blockIdx.y, tid = threadIdx.x;
    if (only_if && only_if[b] != only_if_value) return;
    const int N = n + m;
    const double *Q_ = Qd + (size_t)b * n * n;
    const double *A_ = Ad + (size_t)b * m * n;
    double *Mo = Mout + (size_t)b * N * N;
    const int j0 = 32 * strip, j1 = j0 + 32 < N ? j0 + 32 : N;
    // columns of the x block
    for (int j = j0; j < (j1 < n ? j1 : n); ++j)
        for (int i = tid; i < N; i += 256)
            Mo[(size_t)j * N + i] = i < n ? Q_[(size_t)j * n + i] : A_[(size_t)j * m + (i - n)];
    // columns of the multiplier block: c in [c0, c1)
    const int c0 = (j0 > n ? j0 : n) - n, c1 = j1 - n;
    if (c1 > c0) {
        __shared__ double tile[32][33];
        const int tx = tid & 31, ty = tid >> 5;                      // 32 x 8 threads
        for (int i0 = 0; i0 < n; i0 += 32) {
            for (int r = ty; r < 32; r += 8) {                       // tile[r][tx] = Ad[c0 + tx][i0 + r], read along c
                const int i = i0 + r, c = c0 + tx;
                tile[r][tx] = (i < n && c < c1) ? A_[(size_t)i * m + c] : 0.0;
            }
            __syncthreads();
            for (int cc = ty; cc < 32; cc += 8) {                    // M[i0 + tx][n + c0 + cc] = -tile[tx][cc], written along i
                const int i = i0 + tx, c = c0 + cc;
                if (i < n && c < c1) Mo[(size_t)(n + c) * N + i] = -tile[tx][cc];
            }
            __syncthreads();
        }
        for (int c = c0; c < c1; ++c)
            for (int i = n + tid; i < N; i += 256) Mo[(size_t)(n + c) * N + i] = 0.0;
    }
    if (strip == 0) {
        const double *R_ = R + (size_t)b * n * p;
        const double *B_ = B + (size_t)b * m * p;
        const double *w_ = w + (size_t)b * (size_t)stride_w;
        const size_t vo = (size_t)b * (size_t)N;
        for (int i = tid; i < N; i += 256) {
            double s_;
            if (i < n) {
                s_ = qd[(size_t)b * n + i];
                for (int k = 0; k < p; ++k) s_ = fma(R_[(size_t)k * n + i], w_[k], s_);
                lout[vo + i] = -__builtin_huge_val(); uout[vo + i] = __builtin_huge_val(); kind_out[vo + i] = QPN_ROW_STD;
            } else {
                const int r = i - n;
                s_ = 0.0;
                for (int k = 0; k < p; ++k) s_ = fma(B_[(size_t)k * m + r], w_[k], s_);
                lout[vo + i] = l[(size_t)b * m + r]; uout[vo + i] = u[(size_t)b * m + r];
                kind_out[vo + i] = QPN_ROW_GAVI;
            }
            qout[vo + i] = s_;
        }
    }
}

// ---- schedule hint: permutation of 0..count-1 by descending pivot count (counting sort, ONE workgroup) ----
// Pivot counts are small integers; bins 0..1023 (larger counts share the last bin).  The order inside a
// bin is whatever the atomics give -- any order is a valid schedule.
// ---- (A6) pool assembly: combine_gavis, src/avi.jl:305-377 (+ convert, :113-128, for the reference form) ----------
// One workgroup per (pool instance, strip of 32 columns of the output).  A pool is `players` nodes of one level; its
// decision variables are the union of theirs (nd positions), everything else is a parameter (p columns).  Inputs are
// the players' rows stacked in pool order (sorted ids, :319): Qd [sn x nd] = Q_i[dvars_i, dec_inds], Qp [sn x p] =
// Q_i[dvars_i, param_inds], qd [sn]; Ad [sm x nd] = M2_i[:, dec_inds], Bp [sm x p], l, u [sm]  (sn = sum n_i, sm = sum m_i),
// all column-major.  xi_owner / xi_dpos [sn]: player and decision position of stacked row t; con_owner [sm].
//   REFERENCE form (what the reference hands to PATH): z = [dec (nd) | xi (sn) | lambda-psi (sm) | slack (sm)],
//     rows [ sum_i xi^i_d = 0 (:356-367) | player rows [Q_i  0  -A_i[:, dvars_i]'] (:335-339) | [A -I] | [0 I 0] (:113-128) ],
//     all STD, bounds [free | free | l..u on the slack block].
//   REDUCED form (disjoint decision sets, sn = nd): the structurally dead xi block (multiplied by 0 at :244), its rows
//     and the slack block are dropped: z = [dec | lambda-psi], rows [player rows at their decision positions | A],
//     kinds [STD x nd | GAVI x sm]  -- the form of the solve kernels.
// Item strides of 0 share an input across the batch; Mout stride 0 writes one shared M (config 3: 1 000 payoff draws
// of one game differ only in q).
struct PoolArgs {
    int32_t batch, form, nd, sn, sm, p;
    const int32_t *xi_owner, *xi_dpos, *con_owner, *dec_src;       // dec_src [nd]: stacked row of decision position r (reduced form)
    const double *Qd, *Qp, *qd, *Ad, *Bp, *l, *u, *w;
    int64_t s_Qd, s_Qp, s_qd, s_Ad, s_Bp, s_lu, s_w;
    double *M, *q, *lo, *hi; uint8_t *kind;
    int64_t s_M;
};

__global__ __launch_bounds__(256) void assemble_pools_kernel(PoolArgs a)
{
    const int b = blockIdx.x, strip = blockIdx.y, tid = threadIdx.x;
    const int nd = a.nd, sn = a.sn, sm = a.sm, p = a.p;
    const bool ref = a.form == QPN_POOL_REFERENCE;
    const int d1 = ref ? nd + sn : nd;
    const int N = ref ? d1 + 2 * sm : nd + sm;
    const double *Qd = a.Qd + (size_t)b * (size_t)a.s_Qd, *Ad = a.Ad + (size_t)b * (size_t)a.s_Ad;
    const bool writeM = a.s_M != 0 || b == 0;
    double *Mo = a.M + (size_t)b * (size_t)a.s_M;
    if (writeM) {
        const int j0 = 32 * strip, j1 = j0 + 32 < N ? j0 + 32 : N;
        for (int j = j0; j < j1; ++j) {
            for (int i = tid; i < N; i += 256) {
                double v = 0.0;
                if (ref) {
                    // column blocks: [dec | xi | lam | slack], row blocks: [top nd | player sn | A-rows sm | I-rows sm]
                    if (i < nd) {                                   // sum_i xi^i_d = 0
                        if (j >= nd && j < nd + sn) v = (a.xi_dpos[j - nd] == i) ? 1.0 : 0.0;
                    } else if (i < d1) {
                        const int t = i - nd;
                        if (j < nd) v = Qd[(size_t)j * sn + t];
                        else if (j >= d1 && j < d1 + sm) {
                            const int k = j - d1;
                            if (a.con_owner[k] == a.xi_owner[t]) v = -Ad[(size_t)a.xi_dpos[t] * sm + k];
                        }
                    } else if (i < d1 + sm) {
                        const int k = i - d1;
                        if (j < nd) v = Ad[(size_t)j * sm + k];
                        else if (j == d1 + sm + k) v = -1.0;
                    } else {
                        const int k = i - d1 - sm;
                        if (j == d1 + k) v = 1.0;
                    }
                } else {
                    if (i < nd) {
                        const int t = a.dec_src[i];
                        if (j < nd) v = Qd[(size_t)j * sn + t];
                        else { const int k = j - nd; if (a.con_owner[k] == a.xi_owner[t]) v = -Ad[(size_t)i * sm + k]; }
                    } else if (j < nd) v = Ad[(size_t)j * sm + (i - nd)];
                }
                Mo[(size_t)j * N + i] = v;
            }
        }
    }
    if (strip == 0) {
        const double *Qp = a.Qp + (size_t)b * (size_t)a.s_Qp, *Bp = a.Bp + (size_t)b * (size_t)a.s_Bp;
        const double *qd = a.qd + (size_t)b * (size_t)a.s_qd;
        const double *lc = a.l + (size_t)b * (size_t)a.s_lu, *uc = a.u + (size_t)b * (size_t)a.s_lu;
        const double *w = a.w + (size_t)b * (size_t)a.s_w;
        const size_t vo = (size_t)b * (size_t)N;
        for (int i = tid; i < N; i += 256) {
            double q = 0.0, lo = -QINF, hi = QINF;
            uint8_t kd = QPN_ROW_STD;
            int t = -1, k = -1;                                       // stacked player row / constraint row behind item row i
            if (ref) {
                if (i >= nd && i < d1) t = i - nd;
                else if (i >= d1 && i < d1 + sm) k = i - d1;
                else if (i >= d1 + sm) { lo = lc[i - d1 - sm]; hi = uc[i - d1 - sm]; }
            } else {
                if (i < nd) t = a.dec_src[i];
                else { k = i - nd; lo = lc[k]; hi = uc[k]; kd = QPN_ROW_GAVI; }
            }
            if (t >= 0) { q = qd[t]; for (int c = 0; c < p; ++c) q = fma(Qp[(size_t)c * sn + t], w[c], q); }
            if (k >= 0) { for (int c = 0; c < p; ++c) q = fma(Bp[(size_t)c * sm + k], w[c], q); }
            a.q[vo + i] = q; a.lo[vo + i] = lo; a.hi[vo + i] = hi; a.kind[vo + i] = kd;
        }
    }
}

// ---- (F1) local pieces: local_piece, src/avi_solutions.jl:400-441 + :491-496, for the per-node GAVI of
// process_solution_graph (src/avi.jl:447-477): z = [x_d (n); lambda (m)], w = x_p (p),
//     rows of the piece  [M N ; I2 0 ; I1 0 ; A B],   M = [Qd -Ad'], N = R, A = [Ad 0], B = B   (:405-408)
// with the bounds of recipe K (one code 1..8 per row of z, :413-432).  One workgroup per piece; item t takes the records of
// node node_of[t] (or t): many recipes of one node share its records.  Output BEFORE simplify (polyhedral, host side):
// Ap [(2N) x (N+p)] column-major, lp / up [2N] with the noisy l > u fix (:437-438), entries <= 1e-8 dropped (:439),
// keep [2N] = find_non_trivial (:384-388: a finite bound and a non-empty row).
__global__ __launch_bounds__(256) void local_pieces_kernel(int32_t batch, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd,
                                                            const double *R, const double *qd, const double *Ad,
                                                            const double *B, const double *l, const double *u,
                                                            const int32_t *node_of, const uint8_t *K, double *Ap,
                                                            double *lp, double *up, uint8_t *keep)
{
    const int t = blockIdx.x, tid = threadIdx.x;
    const int b = node_of ? node_of[t] : t;
    const int N = n + m, rows = 2 * N, cols = N + p;
    if ((unsigned)b >= (unsigned)nodes) {
        // a record index outside 0 .. nodes - 1 (device callers are not checked on the host): the piece comes back EMPTY -- every
        // row dropped (keep = 0), bounds (-inf, +inf), zero coefficients -- and no record is read
        for (int r = tid; r < rows; r += 256) { keep[(size_t)t * rows + r] = 0; lp[(size_t)t * rows + r] = -QINF; up[(size_t)t * rows + r] = QINF; }
        double *Az = Ap + (size_t)t * (size_t)rows * cols;
        for (size_t e = tid; e < (size_t)rows * cols; e += 256) Az[e] = 0.0;
        return;
    }
    const double *Q_ = Qd + (size_t)b * n * n, *A_ = Ad + (size_t)b * m * n, *R_ = R + (size_t)b * n * p, *B_ = B + (size_t)b * m * p;
    double *Ao = Ap + (size_t)t * (size_t)rows * cols;
    __shared__ int s_nz[1024];                                       // non-empty flag per row (rows <= 2 * 512)
    for (int r = tid; r < rows; r += 256) s_nz[r] = 0;
    __syncthreads();
    for (int c = 0; c < cols; ++c) {
        for (int r = tid; r < rows; r += 256) {
            double v = 0.0;
            if (r < n) {
                if (c < n) v = Q_[(size_t)c * n + r];
                else if (c < N) v = -A_[(size_t)r * m + (c - n)];
                else v = R_[(size_t)(c - N) * n + r];
            } else if (r < N) v = (c == r) ? 1.0 : 0.0;
            else if (r < N + n) v = (c == r - N) ? 1.0 : 0.0;
            else {
                const int k = r - N - n;
                if (c < n) v = A_[(size_t)c * m + k];
                else if (c >= N) v = B_[(size_t)(c - N) * m + k];
            }
            if (fabs(v) <= 1e-8) v = 0.0; else s_nz[r] = 1;          // (a lane's own rows only: no race)
            Ao[(size_t)c * rows + r] = v;
        }
    }
    __syncthreads();
    const double inf = QINF;
    for (int i = tid; i < N; i += 256) {
        const int code = K[(size_t)t * N + i];
        double b1, b2, b3, b4;
        if (i < n) {
            const double o = qd[(size_t)b * n + i];
            if (code == 1) { b1 = -o; b2 = inf; b3 = -inf; b4 = -inf; }
            else if (code == 2) { b1 = -o; b2 = -o; b3 = -inf; b4 = inf; }
            else if (code == 3) { b1 = -inf; b2 = -o; b3 = inf; b4 = inf; }
            else { b1 = -inf; b2 = inf; b3 = -inf; b4 = inf; }
        } else {
            const double l2 = l[(size_t)b * m + (i - n)], u2 = u[(size_t)b * m + (i - n)];
            if (code == 5) { b1 = 0.0; b2 = inf; b3 = l2; b4 = l2; }
            else if (code == 6) { b1 = 0.0; b2 = 0.0; b3 = l2; b4 = u2; }
            else if (code == 7) { b1 = -inf; b2 = 0.0; b3 = u2; b4 = u2; }
            else { b1 = -inf; b2 = inf; b3 = l2; b4 = u2; }
        }
        if (b1 > b2) b1 = b2;
        if (b3 > b4) b3 = b4;
        const size_t vo = (size_t)t * rows;
        lp[vo + i] = b1; up[vo + i] = b2; lp[vo + N + i] = b3; up[vo + N + i] = b4;
        keep[vo + i] = (uint8_t)((!isinf(b1) || !isinf(b2)) && s_nz[i]);
        keep[vo + N + i] = (uint8_t)((!isinf(b3) || !isinf(b4)) && s_nz[N + i]);
    }
}

// recipe number first + t of the Cartesian product of the rows' code sets (all_Ks, src/avi_solutions.jl:200-215; row 0 is
// the fastest digit): K[t][i] = the digit-th set bit of mask[i] (+1 = code); an empty set gives code 0 (treated as free)
__global__ __launch_bounds__(256) void recipes_kernel(int32_t N, const uint8_t *mask, long long first, int32_t count, uint8_t *K)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    unsigned long long idx = (unsigned long long)first + (unsigned long long)t;
    for (int i = 0; i < N; ++i) {
        const unsigned mk = mask[i];
        const int radix = __popc(mk);
        int code = 0;
        if (radix > 0) {
            int d = (int)(idx % (unsigned)radix);
            idx /= (unsigned)radix;
            unsigned mm = mk;
            while (d-- > 0) mm &= mm - 1;                            // drop the d lowest set bits
            code = __ffs(mm);                                        // 1-based bit position = code
        }
        K[(size_t)t * N + i] = (uint8_t)code;
    }
}

// `key` (optional, [count] int32, zero at first): an exponentially smoothed pivot count per node in units of 1/32 pivot
// (key <- key - key/32 + p): between sweeps the parameters change and with them a node's pivot count, by about half of
// the spread between nodes on the bench workload; ordering by the smoothed count is worth 2-3 % of the sweep there.
__global__ __launch_bounds__(1024) void order_by_pivots_kernel(const int32_t *pivots, int32_t count, int32_t *order, int32_t *key)
{
    __shared__ int hist[1024];
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    hist[t] = 0;
    __syncthreads();
    // pivots == nullptr: the keys are kept up to date by the solve kernel itself (every sweep): sort by them as they are
    auto bin_of = [&](int i) -> int {
        int p = 0;
        if (pivots) { p = pivots[i]; p = p < 0 ? 0 : (p > 1023 ? 1023 : p); }
        if (key) {
            const int k0 = key[i];
            const int k1 = !pivots ? k0 : (k0 > 0 ? k0 - (k0 >> 5) + p : 32 * p);
            p = k1 >> 3;                                    // quarter pivots (the key counts 1/32 pivots)
            p = p < 0 ? 0 : (p > 1023 ? 1023 : p);
        }
        return 1023 - p;                                    // descending: bin 0 holds the largest counts
    };
    for (int i = t; i < count; i += 1024) atomicAdd(&hist[bin_of(i)], 1);
    __syncthreads();
    // exclusive prefix sum of the 1024 bins: wave scans + a scan of the 16 wave totals
    const int v = hist[t];
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if ((t & 63) >= off) incl += o; }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int wv = 0; wv < (t >> 6); ++wv) base += wsum[wv];
    __syncthreads();
    hist[t] = base + incl - v;                              // start of this bin
    __syncthreads();
    for (int i = t; i < count; i += 1024) {
        order[atomicAdd(&hist[bin_of(i)], 1)] = i;
        if (key && pivots) {
            int p = pivots[i];
            p = p < 0 ? 0 : (p > 1023 ? 1023 : p);
            const int k0 = key[i];
            key[i] = k0 > 0 ? k0 - (k0 >> 5) + p : 32 * p;
        }
    }
}

// ---- per-sweep stop / raise decision: (items not solved, max residual), optionally combined over ranks ----
// ONE workgroup.  With world > 1 the local pair is posted into slot [epoch & 1][rank] of every rank's mailbox
// (fine-grained shared buffers, system-scope release store of the epoch last) and the workgroup then waits until
// all `world` slots of ITS OWN mailbox carry this epoch: an all-gather of 24 bytes and the barrier that orders
// the solve kernel's replica stores (earlier on this stream) against the peers' next reads, in one launch.
// Two slot sets by epoch parity: a rank can be at most one epoch ahead of the slowest reader.  Every wait is
// bounded by `timeout_ticks` of the 100 MHz wall clock; on expiry out[2] = 0, out[3] += 1 and the kernel returns.
struct SweepSlot { double nfail, maxres; unsigned long long epoch, pad; };

__device__ __forceinline__ double wave_max_nan_f64(double v)          // max over the wave; a NaN wins
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = (o > v || o != o) ? o : v; }
    return v;
}

__global__ __launch_bounds__(1024) void sweep_status_kernel(const int32_t *status, const double *resid, int32_t count,
                                                            double *out, int32_t rank, int32_t world,
                                                            SweepBoxes boxes, unsigned long long epoch,
                                                            unsigned long long timeout_ticks)
{
    __shared__ double s_res[16];
    __shared__ int s_bad[16];
    const int t = threadIdx.x;
    int bad = 0;
    double mx = 0.0;
    constexpr int PER = 12;                                            // 24 loads in flight per thread and pass: one
    for (int base = 0; base < count; base += PER * 1024) {            // round trip covers 12 288 items
        int st[PER];
        double r[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = base + k * 1024 + t;
            st[k] = i < count ? status[i] : QPN_SUCCESS;
            r[k] = (resid && i < count) ? resid[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            bad += st[k] != QPN_SUCCESS;
            mx = (r[k] > mx || r[k] != r[k]) ? r[k] : mx;              // a NaN residual sticks
        }
    }
    bad = wave_sum_i32(bad);
    mx = wave_max_nan_f64(mx);
    if ((t & 63) == 0) { s_bad[t >> 6] = bad; s_res[t >> 6] = mx; }
    __syncthreads();
    if (t >= 64) return;
    bad = t < 16 ? s_bad[t] : 0;
    mx = t < 16 ? s_res[t] : 0.0;
    bad = wave_sum_i32(bad);
    mx = wave_max_nan_f64(mx);
    double nfail = (double)bad;
    double ok = 1.0;
    if (world > 1) {
        const int par = (int)(epoch & 1ull);
        if (t < world) {                                               // lane t posts to rank t's mailbox (own included)
            SweepSlot *dst = (SweepSlot *)boxes.box[t] + par * QPN_MAX_RANKS + rank;
            dst->nfail = nfail;
            dst->maxres = mx;
            __hip_atomic_store(&dst->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        double pf = 0.0, pr = 0.0;
        if (t < world) {                                               // lane t waits for rank t's post in the own mailbox
            SweepSlot *src = (SweepSlot *)boxes.box[rank] + par * QPN_MAX_RANKS + t;
            const unsigned long long t0 = wall_clock64();
            bool seen = false;
            for (;;) {
                if (__hip_atomic_load(&src->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch) { seen = true; break; }
                if (wall_clock64() - t0 > timeout_ticks) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (seen) { pf = src->nfail; pr = src->maxres; } else ok = 0.0;
        }
        nfail = wave_sum_f64(pf);
        mx = wave_max_nan_f64(pr);
        ok = __all(ok != 0.0) ? 1.0 : 0.0;
    }
    if (t == 0) { out[0] = nfail; out[1] = mx; out[2] = ok; if (ok == 0.0) out[3] += 1.0; }
}

} // namespace

// Are all Qd blocks of these records bitwise symmetric?  One pass when the records become resident (qpn_nodes_upload /
// qpn_nodes_update); flag[0] is set to 1 by any pair (i, j), (j, i) whose bit patterns differ (-0.0 vs +0.0 counts as different).
__global__ __launch_bounds__(256) void qd_asymmetry_kernel(long long total, int32_t n, const double *Qd, int32_t *flag)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long nn = (long long)n * n;
    const long long b = idx / nn;
    const int32_t e = (int32_t)(idx - b * nn), i = e % n, j = e / n;
    if (i >= j) return;
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(Qd) + b * nn;
    if (q[i + (long long)n * j] != q[j + (long long)n * i]) flag[0] = 1;
}

hipError_t qpn_launch_qd_asymmetry(int32_t batch, int32_t n, const double *Qd, int32_t *flag, hipStream_t stream)
{
    const long long total = (long long)batch * n * n;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(qd_asymmetry_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, total, n, Qd, flag);
    return hipGetLastError();
}

hipError_t qpn_launch_sweep_status(const int32_t *status, const double *resid, int32_t count, double *out,
                                   int32_t rank, int32_t world, const SweepBoxes &boxes, unsigned long long epoch,
                                   unsigned long long timeout_ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(sweep_status_kernel, dim3(1), dim3(1024), 0, stream, status, resid, count, out, rank, world,
                       boxes, epoch, timeout_ticks);
    return hipGetLastError();
}

hipError_t qpn_launch_order_by_pivots(const int32_t *pivots, int32_t count, int32_t *order, hipStream_t stream, int32_t *key)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(order_by_pivots_kernel, dim3(1), dim3(1024), 0, stream, pivots, count, order, key);
    return hipGetLastError();
}

hipError_t qpn_launch_local_pieces(int32_t batch, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd, const double *R,
                                   const double *qd, const double *Ad, const double *B, const double *l, const double *u,
                                   const int32_t *node_of, const uint8_t *K, double *Ap, double *lp, double *up, uint8_t *keep,
                                   hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(local_pieces_kernel, dim3((unsigned)batch), dim3(256), 0, stream, batch, nodes, n, m, p, Qd, R, qd, Ad, B, l, u,
                       node_of, K, Ap, lp, up, keep);
    return hipGetLastError();
}

hipError_t qpn_launch_recipes(int32_t N, const uint8_t *mask, long long first, int32_t count, uint8_t *K, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(recipes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, N, mask, first, count, K);
    return hipGetLastError();
}

hipError_t qpn_launch_assemble_pools(const QpnPoolLaunch &L, hipStream_t stream)
{
    if (L.batch <= 0) return hipSuccess;
    PoolArgs a{};
    a.batch = L.batch; a.form = L.form; a.nd = L.nd; a.sn = L.sn; a.sm = L.sm; a.p = L.p;
    a.xi_owner = L.xi_owner; a.xi_dpos = L.xi_dpos; a.con_owner = L.con_owner; a.dec_src = L.dec_src;
    a.Qd = L.Qd; a.Qp = L.Qp; a.qd = L.qd; a.Ad = L.Ad; a.Bp = L.Bp; a.l = L.l; a.u = L.u; a.w = L.w;
    a.s_Qd = L.s_Qd; a.s_Qp = L.s_Qp; a.s_qd = L.s_qd; a.s_Ad = L.s_Ad; a.s_Bp = L.s_Bp; a.s_lu = L.s_lu; a.s_w = L.s_w;
    a.M = L.M; a.q = L.q; a.lo = L.lo; a.hi = L.hi; a.kind = L.kind; a.s_M = L.s_M;
    const int N = L.form == QPN_POOL_REFERENCE ? L.nd + L.sn + 2 * L.sm : L.nd + L.sm;
    hipLaunchKernelGGL(assemble_pools_kernel, dim3((unsigned)L.batch, (unsigned)((N + 31) / 32)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t qpn_launch_check_avi(int32_t batch, int32_t N, const double *M, int64_t strideM,
                                const double *q, const double *l, const double *u,
                                const uint8_t *kind, int64_t stride_kind, const double *z,
                                double tol, int32_t *degree, double *r, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    const int wpb = 4;
    hipLaunchKernelGGL(check_avi_kernel, dim3((batch + wpb - 1) / wpb), dim3(wpb * WAVE), 0, stream,
                       batch, N, M, strideM, q, l, u, kind, stride_kind, z, tol, degree, r);
    return hipGetLastError();
}

hipError_t qpn_launch_comp_indices(int64_t count, const double *zv, const double *rv,
                                   const double *l, const double *u, double tol, int32_t shift,
                                   uint8_t *mask, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(comp_indices_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, count, zv,
                       rv, l, u, tol, shift, mask);
    return hipGetLastError();
}

hipError_t qpn_launch_assemble_nodes(int32_t batch, int32_t n, int32_t m, int32_t p,
                                     const double *Qd, const double *R, const double *qd,
                                     const double *Ad, const double *B, const double *l,
                                     const double *u, const double *w, int64_t stride_w,
                                     double *Mout, double *qout, double *lout, double *uout,
                                     uint8_t *kind_out, hipStream_t stream, const int32_t *only_if,
                                     int32_t only_if_value)
{
    if (batch <= 0) return hipSuccess;
    if (n + m > 64) {
        hipLaunchKernelGGL(assemble_nodes_wide_kernel, dim3((unsigned)batch, (unsigned)((n + m + 31) / 32)), dim3(256), 0,
                           stream, batch, n, m, p, Qd, R, qd, Ad, B, l, u, w, stride_w, Mout, qout,
                           lout, uout, kind_out, only_if, only_if_value);
        return hipGetLastError();
    }
    const int wpb = 4;
    hipLaunchKernelGGL(assemble_nodes_kernel, dim3((batch + wpb - 1) / wpb), dim3(wpb * WAVE), 0,
                       stream, batch, n, m, p, Qd, R, qd, Ad, B, l, u, w, stride_w, Mout, qout,
                       lout, uout, kind_out, only_if, only_if_value);
    return hipGetLastError();
}
