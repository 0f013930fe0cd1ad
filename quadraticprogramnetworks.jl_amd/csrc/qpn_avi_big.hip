// qpn_avi_big.hip -- box-MCP / GAVI pivotal solver for LARGE items (64 < N <= 1024), gfx950.
//
// Same algorithm and arithmetic as qpn_avi_reg.hip; different mapping for
// items whose dictionary does not fit one wavefront's registers (BASELINE config 5: N = 512, and
// the reference-form pools of config 2: N_ref up to ~120).  ONE 256-thread workgroup per AVI:
//   * the N x (N+1) dictionary lives in an HBM workspace, column-major (2.1 MB at N = 512: L2 /
//     Infinity-Cache resident while the item is being worked on); thread t owns rows t, t+256, ...
//     so every column sweep of the rank-1 update is a coalesced 2 KB segment per instruction;
//   * row vectors (basic values, intervals, ids) sit in registers (<= 4 rows per thread), column
//     vectors in LDS; the scaled pivot row is gathered once per pivot into LDS and broadcast;
//   * pivot selection = per-thread partial + LDS tree reduction over 256 threads; every branch is
//     workgroup-uniform.
// This is the correctness/coverage path for large items; the MFMA-tiled blocked variant is the
// planned replacement.
#include "qpn_internal.h"
#include <cstdlib>

#define QINF __builtin_huge_val()

namespace {

constexpr int TPB = 256;
constexpr int RPT = 4;                 // rows per thread: N <= 1024
constexpr int NMAX = TPB * RPT;

struct BigShared {
    double red[TPB];
    int redi[TPB];
    double bc[8];                      // broadcast slots
    int bci[8];
};

__device__ __forceinline__ double block_max(double v, BigShared &S, int tid)
{
    S.red[tid] = v;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (tid < s) { double o = S.red[tid + s]; if (o > S.red[tid]) S.red[tid] = o; }
        __syncthreads();
    }
    double r = S.red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_min(double v, BigShared &S, int tid)
{
    S.red[tid] = v;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (tid < s) { double o = S.red[tid + s]; if (o < S.red[tid]) S.red[tid] = o; }
        __syncthreads();
    }
    double r = S.red[0];
    __syncthreads();
    return r;
}
// smallest index (row id) among threads' candidates, or INT_MAX
__device__ __forceinline__ int block_min_i(int v, BigShared &S, int tid)
{
    S.redi[tid] = v;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (tid < s) { int o = S.redi[tid + s]; if (o < S.redi[tid]) S.redi[tid] = o; }
        __syncthreads();
    }
    int r = S.redi[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ int block_sum_i(int v, BigShared &S, int tid)
{
    S.redi[tid] = v;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (tid < s) S.redi[tid] += S.redi[tid + s];
        __syncthreads();
    }
    int r = S.redi[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(TPB) void avi_solve_big(AviBatchArgs a, double *ws)
{
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (a.only_if && a.only_if[b] != a.only_if_value) return;
    const int N = a.n_items ? a.n_items[b] : a.N, NC = N + 1;     // per-item size (reduced Schur problems)

    __shared__ BigShared S;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    // dynamic LDS: prow[NC], sl[N], su[N], cnb[NC] (nonbasic values), then ints: sat[N], colvar[NC],
    // elist[8N+8], var->row map is not kept (searched)
    double *prow = reinterpret_cast<double *>(dyn);
    double *sl = prow + ((NC + 1) & ~1);
    double *su = sl + N;
    double *cnb = su + N;
    double *valv = cnb + ((NC + 1) & ~1);              // 2N+1 values by variable id (read-back)
    int *sat = reinterpret_cast<int *>(valv + ((2 * N + 2) & ~1));
    int *colvar = sat + N;
    int *elist = colvar + NC + 1;

    double *T = ws + (size_t)b * (size_t)a.N * (size_t)(a.N + 1);       // T[j*N + i]
    const double *Mg = a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)(a.vec_stride ? a.vec_stride : N);
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;
    const double ptol = a.piv_tol, slack = 1e-10;

    // per-thread rows
    int rowvar[RPT]; double xb[RPT], lo[RPT], hi[RPT];
    bool act[RPT];
#pragma unroll
    for (int s = 0; s < RPT; ++s) { act[s] = tid + TPB * s < N; rowvar[s] = -1; xb[s] = 0.0; lo[s] = -QINF; hi[s] = QINF; }

    // ---- pairs: bounds, kinds, initial nonbasic values (pair k handled by the thread owning row k)
    int n_want_local = 0;
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
        const int k = tid + TPB * s;
        if (!act[s]) continue;
        const double lk = a.l[vo + k], uk = a.u[vo + k];
        const int gk = a.kind ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + k] : 0;
        const bool freek = lk == -QINF && uk == QINF, fixedk = lk == uk;
        int atup = 0; double v0 = 0.0;
        if (gk) { colvar[k] = N + k; rowvar[s] = k; lo[s] = lk; hi[s] = uk; }
        else {
            double z0 = (a.flags & QPN_AVI_FLAG_COLD_START) ? 0.0 : a.z[vo + k];
            if (isnan(z0)) z0 = 0.0;
            if (freek) v0 = z0;
            else {
                if (z0 < lk) z0 = lk;
                if (z0 > uk) z0 = uk;
                if (lk == -QINF) { v0 = uk; atup = 1; }
                else if (uk == QINF) { v0 = lk; }
                else if (uk - z0 < z0 - lk) { v0 = uk; atup = 1; }
                else v0 = lk;
                if (uk == lk) atup = 0;
            }
            colvar[k] = k; rowvar[s] = N + k;
            if (fixedk) { lo[s] = -QINF; hi[s] = QINF; }
            else if (freek) { lo[s] = 0.0; hi[s] = 0.0; }
            else if (atup) { lo[s] = -QINF; hi[s] = 0.0; }
            else { lo[s] = 0.0; hi[s] = QINF; }
        }
        sl[k] = lk; su[k] = uk; sat[k] = atup; cnb[k] = v0;
        xb[s] = a.q[vo + k];
        if ((!gk && freek) || (gk && fixedk)) n_want_local++;
    }
    if (tid == 0) { colvar[N] = 2 * N; cnb[N] = 0.0; }
    __syncthreads();

    // ---- copy M into the workspace dictionary, accumulate xb = q + M z_nb (ascending columns), max|M|
    double mabs = 0.0;
    for (int j = 0; j < N; ++j) {
        const double zj = cnb[j];
#pragma unroll
        for (int s = 0; s < RPT; ++s) {
            if (!act[s]) continue;
            const int i = tid + TPB * s;
            const double v = Mg[(size_t)j * N + i];
            T[(size_t)j * N + i] = v;
            mabs = fmax(mabs, fabs(v));
            if (zj != 0.0) xb[s] = fma(v, zj, xb[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < RPT; ++s) if (act[s]) T[(size_t)N * N + tid + TPB * s] = 0.0;
    const double mscale = block_max(mabs, S, tid);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);

    // ---- elist: qualifying pairs in ascending k (serial prefix by thread 0 over a flag array)
    // flags are derived again from sl/su/kind to keep the order deterministic
    int n_enter = 0;
    if (tid == 0) {
        for (int k = 0; k < N; ++k) {
            const int gk = a.kind ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + k] : 0;
            const bool fr = sl[k] == -QINF && su[k] == QINF, fx = sl[k] == su[k];
            if ((!gk && fr) || (gk && fx)) elist[n_enter++] = gk ? N + k : k;
        }
        S.bci[0] = n_enter;
    }
    __syncthreads();
    n_enter = S.bci[0];
    (void)n_want_local;
    __threadfence_block();
    __syncthreads();

    int pivots = 0;

    // position (column) of a nonbasic variable, or -1: block-wide search
    auto col_of = [&](int v) -> int {
        int best = 0x7fffffff;
        for (int j = tid; j < NC; j += TPB) if (colvar[j] == v) best = j;
        int r = block_min_i(best, S, tid);
        return r == 0x7fffffff ? -1 : r;
    };
    auto interval_of = [&](int v, double &l_, double &h_) {
        if (v == 2 * N) { l_ = 0.0; h_ = QINF; return; }
        if (v < N) { l_ = sl[v]; h_ = su[v]; return; }
        const int k = v - N;
        const double L = sl[k], U = su[k];
        if (L == U) { l_ = -QINF; h_ = QINF; }
        else if (L == -QINF && U == QINF) { l_ = 0.0; h_ = 0.0; }
        else if (sat[k]) { l_ = -QINF; h_ = 0.0; }
        else { l_ = 0.0; h_ = QINF; }
    };
    // value held by row r (owner thread publishes): returns broadcast double
    auto row_bcast = [&](const double *arr, int r) -> double {
        if (tid == (r % TPB)) S.bc[0] = arr[r / TPB];
        __syncthreads();
        double v = S.bc[0];
        __syncthreads();
        return v;
    };
    auto rowvar_bcast = [&](int r) -> int {
        if (tid == (r % TPB)) S.bci[1] = rowvar[r / TPB];
        __syncthreads();
        int v = S.bci[1];
        __syncthreads();
        return v;
    };

    // exchange entering column c (moved by delta) with the basic variable of row r
    auto do_pivot = [&](int r, int c, double delta, double leave_val, const double *cm, double inv) {
        const double enter_val = cnb[c] + delta;
        // scaled pivot row -> LDS (strided gather from the workspace)
        for (int j = tid; j < NC; j += TPB) prow[j] = T[(size_t)j * N + r] * inv;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RPT; ++s) if (act[s]) xb[s] = fma(delta, cm[s], xb[s]);
        for (int j = 0; j < NC; ++j) {
            if (j == c) continue;
            const double pr = prow[j];
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                if (!act[s]) continue;
                const int i = tid + TPB * s;
                const size_t off = (size_t)j * N + i;
                T[off] = (i == r) ? -pr : fma(-cm[s], pr, T[off]);
            }
        }
#pragma unroll
        for (int s = 0; s < RPT; ++s) {
            if (!act[s]) continue;
            const int i = tid + TPB * s;
            T[(size_t)c * N + i] = (i == r) ? inv : cm[s] * inv;
        }
        const int ve = colvar[c];
        int vl = 0;
        double nlo, nhi;
        interval_of(ve, nlo, nhi);
        __syncthreads();
        if (tid == (r % TPB)) {
            const int s = r / TPB;
            vl = rowvar[s];
            S.bci[2] = vl;
            rowvar[s] = ve; xb[s] = enter_val; lo[s] = nlo; hi[s] = nhi;
        }
        __syncthreads();
        vl = S.bci[2];
        if (tid == 0) { colvar[c] = vl; cnb[c] = leave_val; }
        __threadfence_block();
        __syncthreads();
        return vl;
    };

    int budget = 4 * N + 4;
    int stage_ = 0, idx = 0;
    int status = QPN_FAILURE;
    int c = N;
    double sigma = -1.0, self_lim = 0.0;
    double cm[RPT];

    for (;;) {
        int e = 0;
        if (stage_ == 0) {
            if (!(idx < n_enter && budget > 0)) { stage_ = 1; continue; }
            e = elist[idx];
            idx++;
            c = col_of(e);
            if (c < 0 || c == N) continue;
        } else if (stage_ == 1) {
            double viol = 0.0;
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (act[s]) { double v = xb[s] < lo[s] ? lo[s] - xb[s] : (xb[s] > hi[s] ? xb[s] - hi[s] : 0.0); if (v > viol) viol = v; }
            const double theta0 = block_max(viol, S, tid);
            if (theta0 <= a.feas_tol) { status = QPN_SUCCESS; break; }
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                if (!act[s]) continue;
                double cov = 0.0;
                if (xb[s] < lo[s]) {
                    double tgt = lo[s] + (theta0 - (lo[s] - xb[s]));
                    if (hi[s] < QINF) { double mid = 0.5 * (lo[s] + hi[s]); if (tgt > mid) tgt = mid; }
                    cov = (tgt - xb[s]) / theta0; xb[s] = tgt;
                } else if (xb[s] > hi[s]) {
                    double tgt = hi[s] - (theta0 - (xb[s] - hi[s]));
                    if (lo[s] > -QINF) { double mid = 0.5 * (lo[s] + hi[s]); if (tgt < mid) tgt = mid; }
                    cov = (tgt - xb[s]) / theta0; xb[s] = tgt;
                }
                T[(size_t)N * N + tid + TPB * s] = cov;
            }
            if (tid == 0) cnb[N] = theta0;
            c = N; sigma = -1.0; self_lim = theta0;
            status = QPN_MAX_ITERS;
            stage_ = 2;
            __threadfence_block();
            __syncthreads();
            continue;
        } else {
            if (pivots >= max_piv) break;
        }

        // column c of the dictionary, this thread's rows
#pragma unroll
        for (int s = 0; s < RPT; ++s) cm[s] = act[s] ? T[(size_t)c * N + tid + TPB * s] : 0.0;

        int r;
        double delta, leave_val, inv;
        if (stage_ == 0) {
            double avm = 0.0;
#pragma unroll
            for (int s = 0; s < RPT; ++s) if (act[s]) avm = fmax(avm, fabs(cm[s]));
            const double colmax = block_max(avm, S, tid);
            double thresh = 1e-9 * (colmax > 1.0 ? colmax : 1.0);
            r = -1; leave_val = 0.0;
            if (e < N && c == e && rowvar_bcast(e) == N + e) {
                const double ad = fabs(row_bcast(cm, e));
                if (ad >= diag_thr) r = e;
            }
            if (r < 0) {
                // largest must-leave row (ties: lowest index)
                double bestl = -1.0;
#pragma unroll
                for (int s = 0; s < RPT; ++s) {
                    if (!act[s]) continue;
                    const int v = rowvar[s];
                    bool ml = false;
                    if (v < 2 * N) {
                        const int k = v < N ? v : v - N;
                        const bool fr = sl[k] == -QINF && su[k] == QINF, fx = sl[k] == su[k];
                        ml = (v >= N) ? fr : (fx && !fr);
                    }
                    if (ml && fabs(cm[s]) > bestl) bestl = fabs(cm[s]);
                }
                double best = block_max(bestl, S, tid);
                if (best > thresh) {
                    int cand = 0x7fffffff;
#pragma unroll
                    for (int s = 0; s < RPT; ++s) {
                        if (!act[s]) continue;
                        const int v = rowvar[s];
                        bool ml = false;
                        if (v < 2 * N) {
                            const int k = v < N ? v : v - N;
                            const bool fr = sl[k] == -QINF && su[k] == QINF, fx = sl[k] == su[k];
                            ml = (v >= N) ? fr : (fx && !fr);
                        }
                        if (ml && fabs(cm[s]) == best && tid + TPB * s < cand) cand = tid + TPB * s;
                    }
                    r = block_min_i(cand, S, tid);
                    const int v = rowvar_bcast(r);
                    leave_val = v >= N ? 0.0 : sl[v];
                } else {
                    double besto = -1.0;
#pragma unroll
                    for (int s = 0; s < RPT; ++s) {
                        if (!act[s]) continue;
                        const int v = rowvar[s];
                        bool ord = false;
                        if (v < 2 * N) {
                            const int k = v < N ? v : v - N;
                            const bool fr = sl[k] == -QINF && su[k] == QINF, fx = sl[k] == su[k];
                            ord = !fr && !fx;
                        }
                        if (ord && fabs(cm[s]) > besto) besto = fabs(cm[s]);
                    }
                    best = block_max(besto, S, tid);
                    if (!(best > thresh)) continue;
                    int cand = 0x7fffffff;
#pragma unroll
                    for (int s = 0; s < RPT; ++s) {
                        if (!act[s]) continue;
                        const int v = rowvar[s];
                        bool ord = false;
                        if (v < 2 * N) {
                            const int k = v < N ? v : v - N;
                            const bool fr = sl[k] == -QINF && su[k] == QINF, fx = sl[k] == su[k];
                            ord = !fr && !fx;
                        }
                        if (ord && fabs(cm[s]) == best && tid + TPB * s < cand) cand = tid + TPB * s;
                    }
                    r = block_min_i(cand, S, tid);
                    const int v = rowvar_bcast(r);
                    if (v < N) {
                        const double x = row_bcast(xb, r), plo = sl[v], phi = su[v];
                        int au;
                        if (x <= plo) { leave_val = plo; au = 0; }
                        else if (x >= phi) { leave_val = phi; au = 1; }
                        else if (plo == -QINF) { leave_val = phi; au = 1; }
                        else if (phi == QINF) { leave_val = plo; au = 0; }
                        else if (phi - x < x - plo) { leave_val = phi; au = 1; }
                        else { leave_val = plo; au = 0; }
                        if (tid == 0) { sat[v] = au; elist[n_enter] = N + v; }
                    } else {
                        leave_val = 0.0;
                        if (tid == 0) elist[n_enter] = v - N;
                    }
                    n_enter++;
                    __threadfence_block();
                    __syncthreads();
                }
            }
            inv = 1.0 / row_bcast(cm, r);
            delta = (leave_val - row_bcast(xb, r)) * inv;
        } else {
            double dmin = QINF;
            double dl[RPT], rcv[RPT]; bool cnd[RPT], isl[RPT];
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                cnd[s] = false; dl[s] = 0.0; rcv[s] = 0.0; isl[s] = false;
                if (!act[s]) continue;
                const double g = sigma * cm[s];
                const double rc = 1.0 / g;
                const bool cl = g < -ptol && lo[s] > -QINF, ch = g > ptol && hi[s] < QINF;
                if (cl || ch) {
                    const double arc = cl ? -rc : rc;
                    const double d = (cl ? xb[s] - lo[s] : hi[s] - xb[s]) * arc;
                    const double d1 = d + slack * arc;
                    cnd[s] = true; dl[s] = d; rcv[s] = rc; isl[s] = cl;
                    if (d1 < dmin) dmin = d1;
                }
            }
            double dmax = block_min(dmin, S, tid);
            if (self_lim < dmax) dmax = self_lim;
            if (dmax == QINF) { status = QPN_RAY_TERM; break; }
            int ncand = 0;
            double bestl = -1.0;
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                if (cnd[s] && dl[s] > dmax) cnd[s] = false;
                if (cnd[s]) {
                    ncand++;
                    double ag = fabs(sigma * cm[s]);
                    if (rowvar[s] == 2 * N) ag = QINF;
                    if (ag > bestl) bestl = ag;
                }
            }
            const int tot = block_sum_i(ncand, S, tid);
            if (tot == 0) {
                const double dlt = sigma * self_lim;
#pragma unroll
                for (int s = 0; s < RPT; ++s) if (act[s]) xb[s] = fma(dlt, cm[s], xb[s]);
                const int ve = colvar[c];
                if (ve == 2 * N) { __syncthreads(); if (tid == 0) cnb[c] = 0.0; status = QPN_SUCCESS; break; }
                const int k = ve;
                const int au = sigma > 0.0 ? 1 : 0;
                const double nv = au ? su[k] : sl[k];
                __syncthreads();
                if (tid == 0) { sat[k] = au; cnb[c] = nv; }
                __threadfence_block();
                __syncthreads();
                pivots++;
                c = col_of(N + k);
                if (c < 0) { status = QPN_FAILURE; break; }
                sigma = au ? -1.0 : 1.0;
                self_lim = QINF;
                continue;
            }
            const double bestg = block_max(bestl, S, tid);
            int cand = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                if (!cnd[s]) continue;
                double ag = fabs(sigma * cm[s]);
                if (rowvar[s] == 2 * N) ag = QINF;
                if (ag == bestg && tid + TPB * s < cand) cand = tid + TPB * s;
            }
            r = block_min_i(cand, S, tid);
            if (tid == (r % TPB)) {
                const int s = r / TPB;
                S.bc[1] = dl[s]; S.bc[2] = isl[s] ? lo[s] : hi[s]; S.bc[3] = rcv[s];
            }
            __syncthreads();
            double step = S.bc[1];
            if (step < 0.0) step = 0.0;
            leave_val = S.bc[2];
            inv = sigma * S.bc[3];
            __syncthreads();
            delta = sigma * step;
        }

        const int vl = do_pivot(r, c, delta, leave_val, cm, inv);
        pivots++;

        if (stage_ == 0) {
            budget--;
            if (n_enter >= 8 * N) stage_ = 1;
        } else {
            if (vl == 2 * N) { status = QPN_SUCCESS; break; }
            int vn;
            if (vl < N) {
                const int k = vl;
                const double Lk = sl[k], Uk = su[k];
                int au = sat[k];
                __syncthreads();
                if (Lk != Uk) { au = (leave_val == Uk) ? 1 : 0; if (tid == 0) sat[k] = au; }
                vn = N + k;
                sigma = au ? -1.0 : 1.0;
                self_lim = QINF;
            } else {
                const int k = vl - N;
                const double Lk = sl[k], Uk = su[k];
                vn = k;
                sigma = sat[k] ? -1.0 : 1.0;
                self_lim = Uk - Lk;
                if (Lk == -QINF && Uk == QINF) { self_lim = QINF; sigma = 1.0; }
            }
            __threadfence_block();
            __syncthreads();
            c = col_of(vn);
            if (c < 0) { status = QPN_FAILURE; break; }
        }
    }

    // ---- read the point back: values by variable id ------------------------------------------------
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RPT; ++s) if (act[s] && rowvar[s] >= 0) valv[rowvar[s]] = xb[s];
    for (int j = tid; j < NC; j += TPB) if (colvar[j] >= 0) valv[colvar[j]] = cnb[j];
    __syncthreads();
    // z into prow (broadcast source for the post-check)
    for (int k = tid; k < N; k += TPB) {
        const int gk = a.kind ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + k] : 0;
        prow[k] = valv[gk ? N + k : k];
    }
    __syncthreads();

    // ---- post-check on the ORIGINAL blocks -------------------------------------------------------------
    double rk[RPT];
#pragma unroll
    for (int s = 0; s < RPT; ++s) rk[s] = act[s] ? a.q[vo + tid + TPB * s] : 0.0;
    for (int j = 0; j < N; ++j) {
        const double zj = prow[j];
        if (zj == 0.0) continue;
#pragma unroll
        for (int s = 0; s < RPT; ++s) if (act[s]) rk[s] = fma(Mg[(size_t)j * N + tid + TPB * s], zj, rk[s]);
    }
    int bad = 0;
    double nres = 0.0;
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
        if (!act[s]) continue;
        const int k = tid + TPB * s;
        const int gk = a.kind ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + k] : 0;
        const double zk = prow[k], lk = sl[k], uk = su[k];
        const double p = gk ? rk[s] : zk, d = gk ? zk : rk[s];
        const double tol = a.check_tol;
        if (d > tol && fabs(p - lk) > tol) bad++;
        if (d < -tol && fabs(p - uk) > tol) bad++;
        if (p - lk < -tol) bad++;
        if (p - uk > tol) bad++;
        if (isnan(p) || isnan(d)) bad++;
        double tt = p - d;
        if (tt < lk) tt = lk;
        if (tt > uk) tt = uk;
        double e = fabs(p - tt);
        if (isnan(e)) e = QINF;
        if (e > nres) nres = e;
        unsigned mask = 0;
        const double ct = a.comp_tol;
        auto approx = [&](double x, double y) { return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= ct); };
        if (!approx(lk, uk)) {
            if (approx(p, lk) && d >= -ct) mask |= 1u;
            if (lk - ct <= p && p <= uk + ct && fabs(d) <= ct) mask |= 2u;
            if (approx(p, uk) && d <= ct) mask |= 4u;
        } else mask = 8u;
        if (gk) mask <<= 4;
        a.z[vo + k] = zk;
        if (a.active) a.active[vo + k] = (uint8_t)mask;
    }
    const int badt = block_sum_i(bad, S, tid);
    const double nrest = block_max(nres, S, tid);
    if (badt > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
    if (tid == 0) {
        a.status[b] = status;
        if (a.resid) a.resid[b] = nrest;
        if (a.pivots) a.pivots[b] = pivots;
    }
}

} // namespace

int qpn_avi_big_max_n() { return NMAX; }

size_t qpn_avi_big_workspace_bytes(int batch, int N)
{
    // dictionary of the general kernel + the views of the blocked-crash path + a kind vector of ones
    return (size_t)batch * (size_t)N * (size_t)(N + 1) * sizeof(double) + qpn_schur_big_workspace_bytes(batch, N) +
           (((size_t)N + 255) & ~(size_t)255);
}

namespace {
__global__ void fill_ones_kernel(uint8_t *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1; }

hipError_t launch_big_kernel(const AviBatchArgs &a, double *dict, hipStream_t stream)
{
    const int N = a.N, NC = N + 1;
    size_t dbl = (size_t)((NC + 1) & ~1) * 2 + 2 * (size_t)N + (size_t)((2 * N + 2) & ~1);
    size_t bytes = dbl * sizeof(double) + sizeof(int) * ((size_t)N + NC + 1 + 8 * (size_t)N + 8);
    bytes = (bytes + 15) & ~(size_t)15;
    static QpnPerDeviceOnce attr_once;
    const int attr_dev = attr_once.device();
    if (!attr_once.done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(avi_solve_big),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_once.done[attr_dev] = true;
    }
    hipLaunchKernelGGL(avi_solve_big, dim3((unsigned)a.batch), dim3(TPB), bytes, stream, a, dict);
    return hipGetLastError();
}
} // namespace

// Large items.  Node-shaped ones (a kind vector is given) first try the blocked MFMA crash of
// qpn_avi_schur_big.hip: stage A -> Lemke on the m x m Schur problem (this kernel, per-item sizes) -> finish;
// what it declines (status -1), and everything else, runs on the general kernel.
hipError_t qpn_launch_avi_solve_big(const AviBatchArgs &a, double *workspace, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    const int N = a.N;
    static const bool no_schur = [] { const char *e = QPN_DEV_ENV("QPN_AVI_BIG_KERNEL"); return e && e[0] == 'g'; }();   // "general"
    if (a.kind == nullptr || a.only_if != nullptr || a.n_items != nullptr || no_schur) return launch_big_kernel(a, workspace, stream);
    double *dict = workspace;
    void *sb = workspace + (size_t)a.batch * (size_t)N * (size_t)(N + 1);
    uint8_t *ones = reinterpret_cast<uint8_t *>(static_cast<char *>(sb) + qpn_schur_big_workspace_bytes(a.batch, N));
    SchurBigWs w{};
    static const bool lemke_general = [] { const char *e = QPN_DEV_ENV("QPN_AVI_BIG_LEMKE"); return e && e[0] == 'g'; }();
    // node path with m <= 64 (uniform, known here): the Schur problems go to the one-wavefront register kernel
    const bool lemke_reg = !lemke_general && a.nd.Qd && a.nd.m >= 1 && a.nd.m <= 64 && a.nd.n + a.nd.m == N &&
                           (a.max_pivots <= 0 || a.max_pivots - a.nd.n >= 1);
    hipError_t e = qpn_launch_schur_big_stage_a(a, sb, &w, !(lemke_general || lemke_reg), stream);
    if (e != hipSuccess) return e;
    if (lemke_general) {
        // A/B path: the general large-item kernel on the Schur problem (per-item sizes), one full dictionary pass per pivot
        hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, ones, N);
        AviBatchArgs r{};
        r.batch = a.batch; r.N = N; r.n_items = w.nred; r.vec_stride = N;
        r.M = w.S; r.strideM = w.s_stride; r.q = w.c; r.l = w.l2; r.u = w.u2; r.kind = ones; r.stride_kind = 0;
        r.z = w.lam; r.status = w.st2; r.pivots = w.piv2; r.resid = nullptr; r.active = nullptr;
        r.check_tol = a.check_tol; r.piv_tol = a.piv_tol; r.feas_tol = a.feas_tol; r.comp_tol = a.comp_tol;
        r.max_pivots = a.max_pivots; r.flags = a.flags | QPN_AVI_FLAG_COLD_START;
        r.only_if = a.status; r.only_if_value = -2;
        e = launch_big_kernel(r, dict, stream);
    } else if (lemke_reg) {
        // the Schur problems are all-GAVI items of size m -- the one-wavefront register kernel solves them (same pivot
        // rule), no workgroup barriers, no dictionary in HBM
        hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, ones, N);
        AviBatchArgs r{};
        r.batch = a.batch; r.N = a.nd.m; r.vec_stride = N;
        r.M = w.S; r.strideM = w.s_stride; r.q = w.c; r.l = w.l2; r.u = w.u2; r.kind = ones; r.stride_kind = 0;
        r.z = w.lam; r.status = w.st2; r.pivots = w.piv2; r.resid = nullptr; r.active = nullptr;
        r.check_tol = a.check_tol; r.piv_tol = a.piv_tol; r.feas_tol = a.feas_tol; r.comp_tol = a.comp_tol;
        r.max_pivots = (a.max_pivots > 0 ? a.max_pivots : 50 * N + 100) - a.nd.n;      // the crash pivots count
        r.flags = a.flags | QPN_AVI_FLAG_COLD_START;
        r.only_if = a.status; r.only_if_value = -2;
        e = qpn_launch_avi_solve_reg(r, stream);
    } else {
        e = qpn_launch_schur_big_lemke(a, w, dict, stream);
    }
    if (e != hipSuccess) return e;
    e = qpn_launch_schur_big_finish(a, w, stream);
    if (e != hipSuccess) return e;
    AviBatchArgs g = a;
    g.only_if = a.status; g.only_if_value = -1;
    return launch_big_kernel(g, dict, stream);
}
