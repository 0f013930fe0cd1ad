// qpn_avi_schur_big.hip -- blocked MFMA crash for LARGE node-shaped items (64 < N <= 1024), gfx950.
//
// BASELINE config 5 (n = m = 256, N = 512) and any item of shape
//     kinds = [STD free x n | GAVI x m],  16 <= ... n <= 512,  M = [[H, C],[A, D]],  q = [g; b].
// The general large-item kernel (qpn_avi_big.hip) keeps the N x (N+1) dictionary in HBM and streams ALL of
// it through the CU once per pivot: n crash pivots + the Lemke pivots, ~2 N^2 * 8 B each.  Here the crash is
// the same construction as in qpn_avi_schur.hip, at workgroup scale:
//   stage_a  (one 256-thread workgroup = 4 wavefronts per item)
//     * only the top half [H | C | g] is eliminated, row-major in an HBM workspace, as 16 x 16 tiles in the
//       C/D layout of v_mfma_f64_16x16x4_f64;
//     * rank-16 block pivots: the 16 pivot columns (a panel, n x 16) go to registers/LDS once, the 16 x 16
//       pivot block is LU-factored and inverted in LDS (no pivoting; every pivot must pass the
//       |u_ss| >= 1e-4 max(1,max|M|) test, else the item is declined), every thread forms its panel row of
//       U' = U P^-1;
//       then T -= U' V with the RAW pivot rows V as B operands: 4 MFMAs per tile and pass, the whole top half
//       is read and written 16 times in all (n / 16 passes) instead of n times;
//     * S = D - A W and c = b - A h as a tiled GEMM on the matrix cores (A operands straight from M).
//   lemke    Lemke on the m x m Schur problem (all GAVI rows) with DELAYED rank-1 updates: a pivot touches one
//            column and one row of the HBM dictionary, the pending pairs are folded in by a rank-8/16 MFMA pass;
//   finish   x = -(W lambda + h), post-check / residual / active sets on the ORIGINAL blocks.
// Declined items (other shape, equality GAVI rows, n > 512, a pivot below the threshold) keep status -1 and
// are solved by the general kernel in a gated launch.  Results: same bar as the small MFMA kernel
// (summation order differs from the scalar crash; primals to ~1e-12, masks and pivot counts identical).
#include "qpn_internal.h"
#include "qpn_tile_chol.h"

#define QINF __builtin_huge_val()

namespace {

constexpr int TPB = 256;
constexpr int PW = 16;                  // panel width (block pivot size)
constexpr int NMAXA = 512;              // rows of the top half held in LDS as U' (512 x 17 doubles = 68 KB)
constexpr int LDU = PW + 1;
__host__ __device__ constexpr int sb_pend_stride(int m) { return ((m + 1 + 15) & ~31) + 16; }

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)

__device__ __forceinline__ int pad16(int v) { return (v + 15) & ~15; }

struct SbShared {
    double red[TPB / 64];    // one slot per wave (sb_block_*)
    int redi[TPB / 64];
    double P[PW * LDU];      // pivot block, then its LU factors (unit lower L below, U on and above the diagonal)
    double rd[PW];           // reciprocals of the pivots
    double Pinv[PW * LDU];   // P^-1 (column c solved by lane c of wave 0)
    int flag;
};

struct SbRed {              // the reduction slots alone (the Lemke kernel: its LDS budget is two workgroups per CU)
    double red[TPB / 64];
    int redi[TPB / 64];
};

// Block reductions (TPB = 256: 4 waves): a shuffle reduction inside each wave, the 4 wave results through LDS --
// two barriers per call (publish / reuse) instead of the ten of a tree over LDS.  NaN never wins a max.
template <class SH> __device__ __forceinline__ double sb_block_max(double v, SH &S, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    if ((tid & 63) == 0) S.red[tid >> 6] = v;
    __syncthreads();
    double r = S.red[0];
#pragma unroll
    for (int k = 1; k < TPB / 64; ++k) { const double o = S.red[k]; r = o > r ? o : r; }
    __syncthreads();
    return r;
}
template <class SH> __device__ __forceinline__ int sb_block_min_i(int v, SH &S, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    if ((tid & 63) == 0) S.redi[tid >> 6] = v;
    __syncthreads();
    int r = S.redi[0];
#pragma unroll
    for (int k = 1; k < TPB / 64; ++k) { const int o = S.redi[k]; r = o < r ? o : r; }
    __syncthreads();
    return r;
}
template <class SH> __device__ __forceinline__ int sb_block_sum_i(int v, SH &S, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) S.redi[tid >> 6] = v;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int k = 0; k < TPB / 64; ++k) r += S.redi[k];
    __syncthreads();
    return r;
}

// Row choice of the ratio test in ONE reduction: the candidate with the largest key, ties -> the lowest row;
// key < 0 = no candidate (returns -1).  Shuffle tournament inside each wave, the 4 wave winners through LDS.
template <class SH> __device__ __forceinline__ int sb_block_argbest(double key, int row, SH &S, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok_ = __shfl_xor(key, off, 64);
        const int or_ = __shfl_xor(row, off, 64);
        if (ok_ > key || (ok_ == key && or_ < row)) { key = ok_; row = or_; }
    }
    if ((tid & 63) == 0) { S.red[tid >> 6] = key; S.redi[tid >> 6] = row; }
    __syncthreads();
    double bk = S.red[0];
    int br = S.redi[0];
#pragma unroll
    for (int k = 1; k < TPB / 64; ++k) {
        const double ok_ = S.red[k];
        const int or_ = S.redi[k];
        if (ok_ > bk || (ok_ == bk && or_ < br)) { bk = ok_; br = or_; }
    }
    __syncthreads();
    return bk < 0.0 ? -1 : br;
}

// ---- stage A ---------------------------------------------------------------------------------------------
// Outputs per item (workspace): Tt (top half after elimination: W = columns n_pad.., h = column n_pad+m_pad),
// S (m x m, column-major), c (m), reduced bounds / start, nsplit = n.  status: -2 accepted, -1 declined.
// LDS_TT: the whole top half lives in LDS behind the U' panels (small nodes whose sizes the host knows: the node
// path) instead of the HBM workspace -- every panel step is then a chain of LDS trips, not of HBM round trips; the
// finished top half is copied to the workspace once, for the finish kernel.
template <bool LDS_TT>
__global__ __launch_bounds__(TPB, 2) void schur_big_stage_a(AviBatchArgs a, SchurBigWs w, int lds_rows, int tt_off, int tt_cap)
{
    const int N = a.N;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    __shared__ SbShared S;
    extern __shared__ __attribute__((aligned(16))) double sUp[];       // U' = U P^-1, [row][LDU]

    const double *Mg = a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)N;

    // ---- shape: leading free STD rows, then GAVI rows with l < u ---------------------------------------
    int first_nonfree = N;
    bool bad_tail = false;
    for (int k = tid; k < N; k += TPB) {
        const double lk = a.l[vo + k], uk = a.u[vo + k];
        const int gk = a.kind ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + k] : 0;
        const bool isfree = !gk && lk == -QINF && uk == QINF;
        if (!isfree && k < first_nonfree) first_nonfree = k;
    }
    const int n = sb_block_min_i(first_nonfree, S, tid);
    const int m = N - n;
    for (int k = tid; k < N; k += TPB) {
        if (k < n) continue;
        const double lk = a.l[vo + k], uk = a.u[vo + k];
        const int gk = a.kind ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + k] : 0;
        if (!gk || lk == uk) bad_tail = true;
    }
    const int nbad = sb_block_sum_i(bad_tail ? 1 : 0, S, tid);
    if (nbad > 0 || n < 1 || m < 1 || n > NMAXA || m > 2 * TPB) { if (tid == 0) a.status[b] = -1; return; }   // (Lemke kernel: 2 rows per thread)

    const int n_pad = pad16(n), m_pad = pad16(m);
    const int ldc = n_pad + m_pad + 16;                 // H part | C part | one tile whose column 0 is g
    const int xcol = n_pad + m_pad;
    double *const Ttg = w.Tt + (size_t)b * (size_t)w.tt_stride;
    double *Tt;
    if constexpr (LDS_TT) {
        if (n_pad * ldc > tt_cap) { if (tid == 0) a.status[b] = -1; return; }     // not the sizes the launch was made for
        Tt = sUp + tt_off;
    } else Tt = Ttg;

    // ---- fill the top half (row-major) and take max |M| over the WHOLE item ------------------------------
    double mabs = 0.0;
    if constexpr (LDS_TT) {
        for (int idx = tid; idx < n_pad * ldc; idx += TPB) {
            const int i = idx / ldc, j = idx - i * ldc;
            Tt[idx] = (i >= n && j == i) ? 1.0 : 0.0;
        }
        __syncthreads();
        // all N x N entries spread over the 256 threads (a column is only N <= ~100 long here), eight loads in flight each
        for (int e0 = 0; e0 < N * N; e0 += 8 * TPB) {
            double v[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { const int e = e0 + q8 * TPB + tid; v[q8] = e < N * N ? Mg[e] : 0.0; }   // coalesced
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                const int e = e0 + q8 * TPB + tid;
                const int j = e / N, i = e - j * N;
                mabs = fmax(mabs, fabs(v[q8]));
                if (e < N * N && i < n) Tt[(size_t)i * ldc + (j < n ? j : n_pad + (j - n))] = v[q8];
            }
        }
    } else {
        // M is column-major, the workspace row-major: a transpose.  Zeros / identity padding first, coalesced along
        // the rows of the workspace; then the top half in 32 x 32 tiles through LDS (the U' area is free here), read
        // along the columns of M and written along the rows of the workspace -- both sides in 256-byte runs (the
        // element-wise version wrote every double of a 1 MB block to its own cache line: 1.8 of stage A's 4.0 ms).
        for (int idx = tid; idx < n_pad * ldc; idx += TPB) {
            const int i = idx / ldc, j = idx - i * ldc;
            Tt[idx] = (i >= n && j == i) ? 1.0 : 0.0;
        }
        __syncthreads();
        double *const tile = sUp;                               // 4 tiles of [32][33]: 16 loads per thread in flight
        const int tx = tid & 31, ty = tid >> 5;
        for (int i0 = 0; i0 < n; i0 += 32)
            for (int j0 = 0; j0 < N; j0 += 128) {
                double v[4][4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {            // tile q4 [col r][row tx] <- M[i0 + tx][j0 + 32 q4 + r]
                        const int i = i0 + tx, j = j0 + 32 * q4 + ty + 8 * rr;
                        v[q4][rr] = (i < N && j < N) ? Mg[(size_t)j * N + i] : 0.0;
                    }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (i0 + tx < n) mabs = fmax(mabs, fabs(v[q4][rr]));    // rows >= n are covered by the sweep below
                        tile[q4 * 1056 + (ty + 8 * rr) * 33 + tx] = v[q4][rr];
                    }
                __syncthreads();
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {            // workspace row i0 + r, columns j0 + 32 q4 + tx
                        const int r = ty + 8 * rr, i = i0 + r, j = j0 + 32 * q4 + tx;
                        if (i < n && j < N) Tt[(size_t)i * ldc + (j < n ? j : n_pad + (j - n))] = tile[q4 * 1056 + tx * 33 + r];
                    }
                __syncthreads();
            }
        // max |M| over the bottom half too: its (N - n) x N entries spread over all threads, eight loads in flight each
        const int mb_ = N - n;
        for (int e0 = 0; e0 < mb_ * N; e0 += 8 * TPB) {
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                const int e = e0 + q8 * TPB + tid;
                const int j = e / mb_, i = n + (e - j * mb_);
                mv[q8] = e < mb_ * N ? Mg[(size_t)j * N + i] : 0.0;
            }
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mabs = fmax(mabs, fabs(mv[q8]));
        }
    }
    for (int i = tid; i < n; i += TPB) Tt[(size_t)i * ldc + xcol] = a.q[vo + i];
    const double mscale = sb_block_max(mabs, S, tid);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);
    __threadfence_block();
    __syncthreads();

    // ---- rank-16 block pivots, two per pass over the tiles where LDS holds two panels -----------------------
    // Factorization of one panel (pivot block kb): P -> LU -> P^-1 in LDS, U' = U P^-1 into sU.  false = a pivot
    // failed the threshold (uniform over the workgroup).
    const int nrt = n_pad / 16, nct = ldc / 16;
    auto factor_panel = [&](int kb, double *sU) -> bool {
        const int p0 = 16 * kb;
        // Pivot block P (rows / columns p0 .. p0+15) -> LU -> P^-1, all in the registers of wave 0: lane i < 16 holds
        // row i, a pivot row is broadcast with v_readlane (every index below is a compile-time constant after
        // unrolling), lane c then solves L y = e_c and U x = y for column c of P^-1 with the factors broadcast the
        // same way.  No pivoting; every pivot must pass the threshold (else the item is declined).
        if (wave == 0) {
            double prow[PW];
            const int li = lane < PW ? lane : 0;
#pragma unroll
            for (int k = 0; k < PW; ++k) prow[k] = Tt[(size_t)(p0 + li) * ldc + p0 + k];
            bool ok = true;
            double rdv[PW];
#pragma unroll
            for (int s_ = 0; s_ < PW; ++s_) {
                const double piv = readlane_f64(prow[s_], s_);
                ok = ok && (fabs(piv) >= diag_thr);
                const double r = 1.0 / piv;
                rdv[s_] = r;
                const double f = prow[s_] * r;                      // l_is of this lane's row (kept in place below the diagonal)
                const bool below = lane > s_;
                if (below) prow[s_] = f;
#pragma unroll
                for (int j2 = s_ + 1; j2 < PW; ++j2) {
                    const double psj = readlane_f64(prow[j2], s_);
                    if (below) prow[j2] = fma(-f, psj, prow[j2]);
                }
            }
            // column `lane` of P^-1: forward (unit lower L), then backward (U, reciprocals rdv)
            double y[PW];
#pragma unroll
            for (int j2 = 0; j2 < PW; ++j2) {
                double sacc = (j2 == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int i2 = 0; i2 < j2; ++i2) sacc = fma(-readlane_f64(prow[i2], j2), y[i2], sacc);
                y[j2] = sacc;
            }
#pragma unroll
            for (int j2 = PW - 1; j2 >= 0; --j2) {
                double sacc = y[j2];
#pragma unroll
                for (int i2 = j2 + 1; i2 < PW; ++i2) sacc = fma(-readlane_f64(prow[i2], j2), y[i2], sacc);
                y[j2] = sacc * rdv[j2];
            }
            if (lane < PW) {
#pragma unroll
                for (int j2 = 0; j2 < PW; ++j2) S.Pinv[j2 * LDU + lane] = y[j2];
            }
            if (lane == 0) S.flag = ok ? 1 : 0;
        }
        __syncthreads();
        if (S.flag == 0) return false;
        // U' = U P^-1, one panel row per thread at a time: 16 accumulators, P^-1 broadcast from LDS (rolled over
        // the panel column j so that nothing but the accumulators stays live); pivot rows carry P - I
        for (int t = tid; t < n_pad; t += TPB) {
            double acc[PW];
#pragma unroll
            for (int k = 0; k < PW; ++k) acc[k] = 0.0;
            const double *ut = Tt + (size_t)t * ldc + p0;
            double uv[PW];                                      // the row's 16 panel entries: one round trip, not 16
#pragma unroll
            for (int j = 0; j < PW; ++j) uv[j] = ut[j];
#pragma unroll 1
            for (int j = 0; j < PW; ++j) {
                double uj = uv[0];
#pragma unroll
                for (int q = 0; q < PW - 1; ++q) uv[q] = uv[q + 1];          // rotate: the rolled loop indexes statically
                if (t == p0 + j) uj -= 1.0;
#pragma unroll
                for (int k = 0; k < PW; ++k) acc[k] = fma(uj, S.Pinv[j * LDU + k], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < PW; ++k) sU[t * LDU + k] = acc[k];
        }
        __syncthreads();
        return true;
    };
    // One pass applies the panels kb and kb + 1 together (half the tile traffic of stage A):
    //     T2 = T0 - (U'_a - U'_b G) V_a(T0) - U'_b V_b(T0),   G = U'_a[rows of block b]  (16 x 16),
    // U'_b being factored from panel b AFTER the update by panel a (a look-ahead update of that one column tile).
    // The pair needs two U' panels in LDS; the launch sizes LDS for one panel of the largest n it accepts, so items
    // with n_pad <= half of that are paired and the others (and an odd last panel) take single steps.
    const bool can_pair = 2 * n_pad <= lds_rows;
    double *const sUa = sUp, *const sUb = sUp + (size_t)n_pad * LDU;
    for (int kb = 0; kb < nrt;) {
        const int p0 = 16 * kb;
        if (!factor_panel(kb, sUa)) { if (tid == 0) a.status[b] = -1; return; }
        const bool pair = can_pair && kb + 1 < nrt;
        if (pair) {
            const int p1 = p0 + 16;
            // look-ahead: column tile kb + 1 gets panel a's update now (16 row tiles over the 4 waves)
            {
                const int J = kb + 1;
                double vb[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) vb[s4] = Tt[(size_t)(p0 + 4 * s4 + lq) * ldc + 16 * J + lc];
                __syncthreads();                                   // every wave has its copy of the raw pivot rows
                for (int I = wave; I < nrt; I += 4) {
                    d4 c0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) c0[g] = Tt[(size_t)(16 * I + 4 * g + lq) * ldc + 16 * J + lc];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) c0 = MFMA(-sUa[(16 * I + lc) * LDU + 4 * s4 + lq], vb[s4], c0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) Tt[(size_t)(16 * I + 4 * g + lq) * ldc + 16 * J + lc] = c0[g];
                }
                __threadfence_block();
                __syncthreads();
            }
            if (!factor_panel(kb + 1, sUb)) { if (tid == 0) a.status[b] = -1; return; }
            // G = U'_a[p1 .. p1+15][:] (kept aside in S.P, free after the factorization), then U'_a -= U'_b G
            if (tid < PW * PW) S.P[(tid / PW) * LDU + (tid % PW)] = sUa[(p1 + tid / PW) * LDU + (tid % PW)];
            __syncthreads();
            for (int t = tid; t < n_pad; t += TPB) {
                double ub_[PW], acc[PW];
#pragma unroll
                for (int q = 0; q < PW; ++q) { ub_[q] = sUb[t * LDU + q]; acc[q] = sUa[t * LDU + q]; }
#pragma unroll 1
                for (int q = 0; q < PW; ++q) {
                    const double uq = ub_[0];
#pragma unroll
                    for (int r = 0; r < PW - 1; ++r) ub_[r] = ub_[r + 1];
#pragma unroll
                    for (int k = 0; k < PW; ++k) acc[k] = fma(-uq, S.P[q * LDU + k], acc[k]);
                }
#pragma unroll
                for (int k = 0; k < PW; ++k) sUa[t * LDU + k] = acc[k];
            }
            __syncthreads();
            // T -= [U~_a | U'_b] [V_a; V_b] on the column tiles right of both panels: 8 k-steps per tile
            for (int J = kb + 2 + wave; J < nct; J += 4) {
                double vb[8];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    vb[s4] = Tt[(size_t)(p0 + 4 * s4 + lq) * ldc + 16 * J + lc];
                    vb[4 + s4] = Tt[(size_t)(p1 + 4 * s4 + lq) * ldc + 16 * J + lc];
                }
                for (int I0 = 0; I0 < nrt; I0 += 2) {
                    const int I1 = I0 + 1 < nrt ? I0 + 1 : I0;
                    d4 c0, c1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        c0[g] = Tt[(size_t)(16 * I0 + 4 * g + lq) * ldc + 16 * J + lc];
                        c1[g] = Tt[(size_t)(16 * I1 + 4 * g + lq) * ldc + 16 * J + lc];
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        c0 = MFMA(-sUa[(16 * I0 + lc) * LDU + 4 * s4 + lq], vb[s4], c0);
                        c0 = MFMA(-sUb[(16 * I0 + lc) * LDU + 4 * s4 + lq], vb[4 + s4], c0);
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        c1 = MFMA(-sUa[(16 * I1 + lc) * LDU + 4 * s4 + lq], vb[s4], c1);
                        c1 = MFMA(-sUb[(16 * I1 + lc) * LDU + 4 * s4 + lq], vb[4 + s4], c1);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) Tt[(size_t)(16 * I0 + 4 * g + lq) * ldc + 16 * J + lc] = c0[g];
                    if (I1 != I0) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) Tt[(size_t)(16 * I1 + 4 * g + lq) * ldc + 16 * J + lc] = c1[g];
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            kb += 2;
            continue;
        }
        // T -= U' V on the live column tiles (those right of the panel), 4 k-steps per tile
        for (int J = kb + 1 + wave; J < nct; J += 4) {
            double vb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) vb[s] = Tt[(size_t)(p0 + 4 * s + lq) * ldc + 16 * J + lc];
            // two row tiles at a time: 8 loads in flight ahead of the MFMAs (the pass is a stream over HBM; four at a
            // time is no faster -- the stream, not its latency, is the limit)
            for (int I0 = 0; I0 < nrt; I0 += 2) {
                const int I1 = I0 + 1 < nrt ? I0 + 1 : I0;
                d4 c0, c1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    c0[g] = Tt[(size_t)(16 * I0 + 4 * g + lq) * ldc + 16 * J + lc];
                    c1[g] = Tt[(size_t)(16 * I1 + 4 * g + lq) * ldc + 16 * J + lc];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) c0 = MFMA(-sUa[(16 * I0 + lc) * LDU + 4 * s + lq], vb[s], c0);
#pragma unroll
                for (int s = 0; s < 4; ++s) c1 = MFMA(-sUa[(16 * I1 + lc) * LDU + 4 * s + lq], vb[s], c1);
#pragma unroll
                for (int g = 0; g < 4; ++g) Tt[(size_t)(16 * I0 + 4 * g + lq) * ldc + 16 * J + lc] = c0[g];
                if (I1 != I0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) Tt[(size_t)(16 * I1 + 4 * g + lq) * ldc + 16 * J + lc] = c1[g];
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        kb += 1;
    }

    // ---- S = D - A W, c = b - A h: tiled GEMM, A operands from the original M ---------------------------
    double *Sg = w.S + (size_t)b * (size_t)w.s_stride;
    double *cg = w.c + vo;
    const int mt = m_pad / 16;
    // Large items (the HBM-workspace variant, with the U' panels' LDS free again): the four waves share ONE 8 x 8
    // super-block of output tiles (a 4 x 4 quadrant each) and the operand slabs of 16 k-values are staged through
    // LDS by the whole workgroup, coalesced -- an operand byte then feeds four waves instead of one: 2.1 MB of
    // operand traffic per item at n = m = 256 instead of 5 MB.  The extra column c = b - A h is a separate pass.
    constexpr int SBK = 16, SBW = 128, SBLD = SBW + 4;               // slab depth, super-block width (rows / cols), padded ld
    const bool staged = !LDS_TT && (size_t)lds_rows * LDU >= (size_t)2 * SBK * SBLD && mt >= 8;
    if (staged) {
        double *const As = sUp, *const Ws = sUp + SBK * SBLD;        // As[k][row], Ws[k][col]
        const int rsb = (mt + 7) / 8, csb = (mt + 7) / 8;
        const int qi = wave >> 1, qj = wave & 1;
        for (int sb = 0; sb < rsb * csb; ++sb) {
            const int I00 = 8 * (sb / csb), J00 = 8 * (sb % csb);
            const int I0 = I00 + 4 * qi, J0 = J00 + 4 * qj;
            d4 acc[4][4];
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ri = 16 * (I0 + bi) + 4 * g + lq, cj = 16 * (J0 + bj) + lc;
                        acc[bi][bj][g] = (ri < m && cj < m) ? Mg[(size_t)(n + cj) * N + n + ri] : 0.0;
                    }
            for (int k0 = 0; k0 < n_pad; k0 += SBK) {
                __syncthreads();                                     // the previous slab has been consumed
                {
                    const int r = tid & (SBW - 1), kq = tid >> 7;    // 256 threads: 128 rows (cols) x 2 k-phases
                    double va[SBK / 2], vw[SBK / 2];
#pragma unroll
                    for (int i2 = 0; i2 < SBK / 2; ++i2) {
                        const int k = k0 + kq + 2 * i2;
                        const int ar = 16 * I00 + r, wc = 16 * J00 + r;
                        const bool oka = k < n && ar < m;
                        va[i2] = oka ? -Mg[(size_t)k * N + n + ar] : 0.0;
                        vw[i2] = (k < n_pad && wc < m_pad) ? Tt[(size_t)k * ldc + n_pad + wc] : 0.0;
                    }
#pragma unroll
                    for (int i2 = 0; i2 < SBK / 2; ++i2) {
                        As[(kq + 2 * i2) * SBLD + r] = va[i2];
                        Ws[(kq + 2 * i2) * SBLD + r] = vw[i2];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < SBK / 4; ++kk) {
                    double av[4], bv[4];
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi) av[bi] = As[(4 * kk + lq) * SBLD + 16 * (4 * qi + bi) + lc];
#pragma unroll
                    for (int bj = 0; bj < 4; ++bj) bv[bj] = Ws[(4 * kk + lq) * SBLD + 16 * (4 * qj + bj) + lc];
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
                        for (int bj = 0; bj < 4; ++bj) acc[bi][bj] = MFMA(av[bi], bv[bj], acc[bi][bj]);
                }
            }
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ri = 16 * (I0 + bi) + 4 * g + lq, cj = 16 * (J0 + bj) + lc;
                        if (ri < m && cj < m) Sg[w.s_rowmajor ? (size_t)ri * m + cj : (size_t)cj * m + ri] = acc[bi][bj][g];
                    }
        }
        // c = b - A h, one row per thread, k ascending (h = column xcol of the top half)
        __syncthreads();
        for (int k = tid; k < n; k += TPB) As[k] = Tt[(size_t)k * ldc + xcol];
        __syncthreads();
        for (int ri = tid; ri < m; ri += TPB) {
            double acc_c = a.q[vo + n + ri];
            int k = 0;
            for (; k + 8 <= n; k += 8) {                              // eight column entries in flight
                double mv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) mv[q8] = Mg[(size_t)(k + q8) * N + n + ri];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) acc_c = fma(-mv[q8], As[k + q8], acc_c);
            }
            for (; k < n; ++k) acc_c = fma(-Mg[(size_t)k * N + n + ri], As[k], acc_c);
            cg[ri] = acc_c;
        }
    } else {
    // Register-blocked: a wave owns a block of GB x GB output tiles and streams k -- per k-step GB + GB operand
    // loads feed GB * GB MFMAs (one operand load per 2 MFMAs instead of 2 per MFMA), two k-steps of operands in
    // flight.  Tiles outside the item multiply zeros on clamped addresses and are not stored.
    constexpr int GB = 4;
    const int rbk = (mt + GB - 1) / GB, cbk = (mt + 1 + GB - 1) / GB;
    for (int t = wave; t < rbk * cbk; t += 4) {
        const int I0 = GB * (t / cbk), J0 = GB * (t % cbk);
        d4 acc[GB][GB];
#pragma unroll
        for (int bi = 0; bi < GB; ++bi)
#pragma unroll
            for (int bj = 0; bj < GB; ++bj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int I = I0 + bi, J = J0 + bj;
                    const int ri = 16 * I + 4 * g + lq, cj = 16 * J + lc;
                    double v = 0.0;
                    if (I < mt && J < mt) { if (ri < m && cj < m) v = Mg[(size_t)(n + cj) * N + n + ri]; }
                    else if (I < mt && J == mt && lc == 0 && ri < m) v = a.q[vo + n + ri];
                    acc[bi][bj][g] = v;
                }
        auto load_ops = [&](int kk, double (&av)[GB], double (&bv)[GB]) {
            const bool okk = kk < n_pad / 4;
            const int aj = okk ? 4 * kk + lq : 0;
#pragma unroll
            for (int bi = 0; bi < GB; ++bi) {
                const int ai = 16 * (I0 + bi) + lc;
                const bool oka = okk && ai < m && aj < n;
                const double ta = Mg[(size_t)(oka ? aj : 0) * N + n + (oka ? ai : 0)];
                av[bi] = oka ? -ta : 0.0;
            }
#pragma unroll
            for (int bj = 0; bj < GB; ++bj) {
                const bool okb = okk && J0 + bj <= mt;
                const double tb = Tt[(size_t)aj * ldc + n_pad + 16 * (okb ? J0 + bj : 0) + lc];
                bv[bj] = okb ? tb : 0.0;
            }
        };
        double a0[GB], b0[GB], a1[GB], b1[GB];
        load_ops(0, a0, b0);
        for (int kk = 0; kk < n_pad / 4; kk += 2) {           // n_pad / 4 is even
            load_ops(kk + 1, a1, b1);
#pragma unroll
            for (int bi = 0; bi < GB; ++bi)
#pragma unroll
                for (int bj = 0; bj < GB; ++bj) acc[bi][bj] = MFMA(a0[bi], b0[bj], acc[bi][bj]);
            load_ops(kk + 2, a0, b0);                          // past the end: zeros on clamped addresses
#pragma unroll
            for (int bi = 0; bi < GB; ++bi)
#pragma unroll
                for (int bj = 0; bj < GB; ++bj) acc[bi][bj] = MFMA(a1[bi], b1[bj], acc[bi][bj]);
        }
#pragma unroll
        for (int bi = 0; bi < GB; ++bi)
#pragma unroll
            for (int bj = 0; bj < GB; ++bj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int I = I0 + bi, J = J0 + bj;
                    const int ri = 16 * I + 4 * g + lq, cj = 16 * J + lc;
                    if (I < mt && J < mt) { if (ri < m && cj < m) Sg[w.s_rowmajor ? (size_t)ri * m + cj : (size_t)cj * m + ri] = acc[bi][bj][g]; }
                    else if (I < mt && J == mt && lc == 0 && ri < m) cg[ri] = acc[bi][bj][g];
                }
    }
    }
    if constexpr (LDS_TT) {
        __syncthreads();
        for (int idx = tid; idx < n_pad * ldc; idx += TPB) Ttg[idx] = Tt[idx];
    }
    // reduced problem data: bounds of the GAVI rows, cold start
    for (int k = tid; k < m; k += TPB) {
        w.l2[vo + k] = a.l[vo + n + k]; w.u2[vo + k] = a.u[vo + n + k]; w.lam[vo + k] = 0.0;
    }
    if (tid == 0) { w.nsplit[b] = n; w.nred[b] = m; a.status[b] = -2; }
}

// ---- finish: x = -(W lambda + h), post-check on the ORIGINAL blocks --------------------------------------
__global__ __launch_bounds__(TPB) void schur_big_finish(AviBatchArgs a, SchurBigWs w)
{
    const int N = a.N;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (a.status[b] != -2) return;
    __shared__ SbShared S;
    extern __shared__ __attribute__((aligned(16))) double zs[];       // z, N doubles
    const int n = w.nsplit[b], m = N - n;
    const int n_pad = pad16(n), m_pad = pad16(m);
    const int ldc = n_pad + m_pad + 16, xcol = n_pad + m_pad;
    const double *Tt = w.Tt + (size_t)b * (size_t)w.tt_stride;
    const double *Mg = a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)N;
    for (int k = tid; k < m; k += TPB) zs[n + k] = w.lam[vo + k];
    __syncthreads();
    for (int i = tid; i < n; i += TPB) {
        double s = Tt[(size_t)i * ldc + xcol];
        const double *wr = Tt + (size_t)i * ldc + n_pad;
        for (int k = 0; k < m; ++k) s = fma(wr[k], zs[n + k], s);
        zs[i] = -s;
    }
    __syncthreads();
    int bad = 0;
    double nres = 0.0;
    for (int k = tid; k < N; k += TPB) {
        double rk = a.q[vo + k];
        int j = 0;
        for (; j + 8 <= N; j += 8) {                       // eight column entries in flight; a zero z_j contributes nothing
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mv[q8] = Mg[(size_t)(j + q8) * N + k];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { const double zj = zs[j + q8]; rk = (zj != 0.0) ? fma(mv[q8], zj, rk) : rk; }
        }
        for (; j < N; ++j) {
            const double zj = zs[j];
            if (zj != 0.0) rk = fma(Mg[(size_t)j * N + k], zj, rk);
        }
        const int gk = k >= n;
        const double zk = zs[k], lk = a.l[vo + k], uk = a.u[vo + k];
        const double p = gk ? rk : zk, d = gk ? zk : rk;
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
    const int badt = sb_block_sum_i(bad, S, tid);
    const double nrest = sb_block_max(nres, S, tid);
    if (tid == 0) {
        int status = w.st2[b];
        if (badt > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
        a.status[b] = status;
        if (a.resid) a.resid[b] = nrest;
        if (a.pivots) a.pivots[b] = n + w.piv2[b];
    }
}


// ---- Lemke on the m x m Schur problem with DELAYED updates ------------------------------------------------
// All pairs are GAVI pairs: p_k = s_k = (S lambda + c)_k in [l_k, u_k], d_k = lambda_k (ids k, m + k; artificial
// 2m; column m of the dictionary = the extra / covering column).  Same rule as Stage B of qpn_avi_schur.hip.
// A Gauss-Jordan exchange on (r, c) with pivot p, column u, row w IS a rank-1 update
//         T' = T - (u + e_r)(w - e_c)' / p,
// so the dictionary is kept as  T = T_base - sum_k a_k b_k'  with T_base in HBM (row-major) and up to KP pending
// (a_k, b_k) pairs in LDS.  A pivot then touches ONE column and ONE row of T_base (2 + 8 KB) instead of all of
// it (1 MB read + written at m = 256); every KP pivots the pending pairs are folded into T_base with a rank-KP
// MFMA pass.  Entries of the exchanged row / column carry a relative error ~ eps |p| (cancellation in the
// rank-1 form); the result is certified by the post-check on the original blocks like every other path.
#ifdef QPN_STAMPS
#define LSTAMP(slot)                                                    \
    do {                                                                \
        unsigned long long now__ = __builtin_amdgcn_s_memtime();        \
        __builtin_amdgcn_s_waitcnt(0xC07F);                             \
        stamp_acc[slot] += now__ - stamp_last;                          \
        stamp_last = now__;                                             \
    } while (0)
#else
#define LSTAMP(slot) do { } while (0)
#endif
template <int KP>
__global__ __launch_bounds__(TPB) void schur_big_lemke(AviBatchArgs a, SchurBigWs w, double *dict, int m_lo, int m_hi, int after_bpp)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    if (a.status[b] != -2) return;
    if (after_bpp && w.st2[b] == QPN_SUCCESS) return;       // schur_big_bpp (below) has finished this node and written lambda
#ifdef QPN_STAMPS
    // (diagnostic builds: thread 0's clocks per phase, second half of a [2][batch][8] buffer -- tools/big2_stamps.py)
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    const int wave = tid >> 6, lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    // (row stride of T_base: a multiple of 16 doubles -- with m + 1 every 16-column run of a row straddles two cache lines and
    // the fold's tile stores are partial-sector writes)
    const int m = w.nred[b], XC = m, VTH = 2 * m;
    // (an item's slot in `dict` is N (N + 1) doubles: with very few free rows -- n (2 m + n + 1) < 15 m, e.g. n = 5, m = 99 -- the
    //  padded rows do not fit it and the plain stride is used; the blocked crash from the records, the only writer that
    //  fills T_base itself (s_rowmajor == 2), has n > 64 and always fits)
    const int ldp = (m + 1 + 15) & ~15;
    const int ld = ((long long)m * ldp <= (long long)a.N * (a.N + 1)) ? ldp : m + 1;
    if (m < m_lo || m > m_hi) return;                // this launch's LDS is sized for m_hi
    const size_t vo = (size_t)b * (size_t)a.N;
    __shared__ SbRed S;
    __shared__ double bcd[8];
    __shared__ int bci[8];
    extern __shared__ __attribute__((aligned(16))) double dynl[];
    // row stride of the pending pairs: >= m + 1 and == 16 mod 32 doubles, so that the four k-rows an MFMA operand of the fold
    // reads (kk = 4 s + lq) sit on different LDS banks (stride m or m + 1: a 4-way / 2-way conflict on every operand read)
    const int mA = sb_pend_stride(m), mB = mA;
    double *PA = dynl;                               // [KP][mA]: m used
    double *PB = PA + (size_t)KP * mA;               // [KP][mB]: m + 1 used
    double *cnb = PB + (size_t)KP * mB;              // nonbasic values by column, [m + 1]
    double *lo0 = cnb + (m + 1) + ((m + 1) & 1);     // pair bounds
    double *hi0 = lo0 + m;
    int *colvar = reinterpret_cast<int *>(hi0 + m);  // [m + 1]
    int *sat = colvar + (m + 1) + ((m + 1) & 1);     // [m]
    double *Tb = dict + (size_t)b * (size_t)a.N * (size_t)(a.N + 1);     // T_base, row-major, m rows of ld
    const double *Sg = w.S + (size_t)b * (size_t)w.s_stride;

    // rows owned by this thread
    bool act[2]; int rowvar[2]; double xb[2], lo[2], hi[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = tid + TPB * h;
        act[h] = i < m; rowvar[h] = -1; xb[h] = 0.0; lo[h] = -QINF; hi[h] = QINF;
        if (act[h]) {
            lo[h] = w.l2[vo + i]; hi[h] = w.u2[vo + i]; xb[h] = w.c[vo + i]; rowvar[h] = i;
            lo0[i] = lo[h]; hi0[i] = hi[h]; sat[i] = 0; colvar[i] = m + i; cnb[i] = 0.0;
        }
    }
    if (tid == 0) { colvar[m] = VTH; cnb[m] = 0.0; }
    // T_base = [S | 0]: S is column-major, read coalesced over rows
    if (w.s_rowmajor == 2) {
        // (stage A wrote S straight into T_base)
    } else if (w.s_rowmajor) {
        for (int idx = tid; idx < m * m; idx += TPB) { const int i = idx / m, j = idx - i * m; Tb[(size_t)i * ld + j] = Sg[idx]; }
    } else {
        for (int j = 0; j < m; ++j)
            for (int i = tid; i < m; i += TPB) Tb[(size_t)i * ld + j] = Sg[(size_t)j * m + i];
    }
    __syncthreads();

    int pivots = 0, status = QPN_FAILURE, npend = 0;
    // the budget counts the n crash pivots of stage A too (`pivots` here counts this kernel's own)
    const int max_piv = (a.max_pivots > 0 ? a.max_pivots : 50 * (a.n_items ? a.n_items[b] : a.N) + 100) - w.nsplit[b];
    int c = XC;
    bool sneg = true;
    double self_lim = 0.0, elo = 0.0, ehi = QINF;
    const double slack = 1e-10, ptol = a.piv_tol;
    {
        double viol = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) if (act[h]) viol = fmax(viol, xb[h] < lo[h] ? lo[h] - xb[h] : (xb[h] > hi[h] ? xb[h] - hi[h] : 0.0));
        const double theta0 = sb_block_max(viol, S, tid);
        if (theta0 <= a.feas_tol) status = QPN_SUCCESS;
        else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!act[h]) continue;
                double cov = 0.0;
                if (xb[h] < lo[h]) {
                    double tgt = lo[h] + (theta0 - (lo[h] - xb[h]));
                    if (hi[h] < QINF) { const double mid = 0.5 * (lo[h] + hi[h]); if (tgt > mid) tgt = mid; }
                    cov = (tgt - xb[h]) / theta0; xb[h] = tgt;
                } else if (xb[h] > hi[h]) {
                    double tgt = hi[h] - (theta0 - (xb[h] - hi[h]));
                    if (lo[h] > -QINF) { const double mid = 0.5 * (lo[h] + hi[h]); if (tgt < mid) tgt = mid; }
                    cov = (tgt - xb[h]) / theta0; xb[h] = tgt;
                }
                Tb[(size_t)(tid + TPB * h) * ld + m] = cov;
            }
            if (tid == 0) cnb[m] = theta0;
            self_lim = theta0;
            status = QPN_MAX_ITERS;
        }
    }
    __threadfence_block();
    __syncthreads();

    auto col_of = [&](int v) -> int {
        int best = 0x7fffffff;
        for (int j = tid; j <= m; j += TPB) if (colvar[j] == v) best = j;
        const int r = sb_block_min_i(best, S, tid);
        return r == 0x7fffffff ? -1 : r;
    };

    LSTAMP(0);   // setup
    while (status == QPN_MAX_ITERS) {
        if (pivots >= max_piv) break;
        // ---- entering column of the CURRENT dictionary: T_base column minus the pending rank-1 terms
        double cm[2] = {0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!act[h]) continue;
            const int i = tid + TPB * h;
            double v = Tb[(size_t)i * ld + c];
            for (int k = 0; k < npend; ++k) v = fma(-PA[k * mA + i], PB[k * mB + c], v);
            cm[h] = v;
        }
        LSTAMP(1);   // entering column (T_base + pending terms)
        // ---- ratio test
        double gdir[2], rc[2], dd[2], d1min = QINF;
        bool cndlo[2], cnd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            gdir[h] = sneg ? -cm[h] : cm[h];
            rc[h] = 1.0 / gdir[h];
            cndlo[h] = act[h] && gdir[h] < -ptol && lo[h] > -QINF;
            const bool cndhi = act[h] && gdir[h] > ptol && hi[h] < QINF;
            cnd[h] = cndlo[h] || cndhi;
            const double arc = fabs(rc[h]);
            dd[h] = (cndlo[h] ? xb[h] - lo[h] : hi[h] - xb[h]) * arc;
            if (cnd[h]) d1min = fmin(d1min, fma(slack, arc, dd[h]));
        }
        double dmax = -sb_block_max(-d1min, S, tid);
        if (self_lim < dmax) dmax = self_lim;
        if (dmax == QINF) { status = QPN_RAY_TERM; break; }
        // candidates, and among them the pivot row (largest |pivot|, the artificial first; ties -> lowest row)
        bool cand[2];
        double ag = -1.0;
        int myr = 0x7fffffff;
#pragma unroll
        for (int h = 1; h >= 0; --h) {
            cand[h] = cnd[h] && dd[h] <= dmax;
            if (cand[h]) {
                const double v = rowvar[h] == VTH ? QINF : fabs(gdir[h]);
                if (v >= ag) { ag = v; myr = tid + TPB * h; }          // h = 0 (the lower row) wins a tie within the thread
            }
        }
        const int r = sb_block_argbest(ag, myr, S, tid);
        LSTAMP(2);   // ratio test + the two reductions
        const int ncand = r < 0 ? 0 : 1;
        if (ncand == 0) {
            // the entering variable reaches its own opposite bound first: no basis change
            const double dl = sneg ? -self_lim : self_lim;
#pragma unroll
            for (int h = 0; h < 2; ++h) if (act[h]) xb[h] = fma(dl, cm[h], xb[h]);
            const int ve = colvar[c];
            if (ve == VTH) { __syncthreads(); if (tid == 0) cnb[c] = 0.0; status = QPN_SUCCESS; break; }
            const int k = ve;
            const int au = sneg ? 0 : 1;
            const double nv = au ? hi0[k] : lo0[k];
            __syncthreads();
            if (tid == 0) { sat[k] = au; cnb[c] = nv; }
            pivots++;
            __syncthreads();
            c = col_of(m + k);
            if (c < 0) { status = QPN_FAILURE; break; }
            sneg = au != 0;
            self_lim = QINF;
            if (au) { elo = -QINF; ehi = 0.0; } else { elo = 0.0; ehi = QINF; }
            continue;
        }
        // owner of row r publishes its scalars
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (tid + TPB * h == r) {
                bcd[0] = dd[h]; bcd[1] = cndlo[h] ? lo[h] : hi[h]; bcd[2] = rc[h]; bcd[3] = cm[h];
                bci[0] = rowvar[h]; bci[1] = cndlo[h] ? 0 : 1;
            }
        }
        __syncthreads();
        double step = bcd[0];
        if (step < 0.0) step = 0.0;
        const double leave_val = bcd[1];
        const double inv = sneg ? -bcd[2] : bcd[2];
        const double delta = sneg ? -step : step;
        const int vl = bci[0], hit_hi = bci[1];
        const double enter_val = cnb[c] + delta;
        const int ve = colvar[c];
        // ---- pivot row of the CURRENT dictionary -> pending pair (a, b) = ((u + e_r) inv, w - e_c)
        for (int j = tid; j <= m; j += TPB) {
            double v = Tb[(size_t)r * ld + j];
            for (int k = 0; k < npend; ++k) v = fma(-PA[k * mA + r], PB[k * mB + j], v);
            PB[npend * mB + j] = v - (j == c ? 1.0 : 0.0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!act[h]) continue;
            const int i = tid + TPB * h;
            PA[npend * mA + i] = (cm[h] + (i == r ? 1.0 : 0.0)) * inv;
            if (i == r) { xb[h] = enter_val; rowvar[h] = ve; lo[h] = elo; hi[h] = ehi; }
            else xb[h] = fma(delta, cm[h], xb[h]);
        }
        __syncthreads();
        LSTAMP(3);   // pivot row (T_base + pending terms) -> pending pair
        if (tid == 0) { colvar[c] = vl; cnb[c] = leave_val; }
        npend++;
        pivots++;
        if (vl == VTH) { status = QPN_SUCCESS; break; }
        int vn;
        {
            const int k = vl < m ? vl : vl - m;
            const double Lk = lo0[k], Uk = hi0[k];
            const bool isfree = Lk == -QINF && Uk == QINF;
            int au = sat[k];
            if (vl < m) {
                au = hit_hi;
                __syncthreads();
                if (tid == 0) sat[k] = au;
                vn = m + k; sneg = au != 0; self_lim = QINF;
                if (isfree) { elo = 0.0; ehi = 0.0; }
                else if (au) { elo = -QINF; ehi = 0.0; }
                else { elo = 0.0; ehi = QINF; }
            } else {
                vn = k; sneg = au != 0; self_lim = Uk - Lk;
                if (isfree) { self_lim = QINF; sneg = false; }
                elo = Lk; ehi = Uk;
            }
        }
        __syncthreads();
        c = col_of(vn);
        LSTAMP(4);   // bookkeeping + the next column's index
        if (c < 0) { status = QPN_FAILURE; break; }
        // ---- fold the pending pairs into T_base: rank-KP MFMA pass over its tiles
        if (npend == KP) {
            // FT tiles per wave in flight: the loads of all of them are issued before the first MFMA, so the pass
            // waits for HBM once per FT tiles instead of once per tile
            constexpr int FT = 4;
            const int rt = (m + 15) / 16, ct = (m + 1 + 15) / 16, nt = rt * ct;
            for (int t0 = wave; t0 < nt; t0 += 4 * FT) {
                d4 acc[FT];
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const int t = t0 + 4 * f;
                    const int I = t / ct, J = t % ct;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ri = 16 * I + 4 * g + lq, cj = 16 * J + lc;
                        acc[f][g] = (t < nt && ri < m && cj <= m) ? Tb[(size_t)ri * ld + cj] : 0.0;
                    }
                }
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const int t = t0 + 4 * f;
                    const int I = t / ct, J = t % ct;
                    if (t >= nt) break;                          // wave-uniform
#pragma unroll
                    for (int s4 = 0; s4 < KP / 4; ++s4) {
                        const int ri = 16 * I + lc, cj = 16 * J + lc, kk = 4 * s4 + lq;
                        const double av = ri < m ? -PA[kk * mA + ri] : 0.0;
                        const double bv = cj <= m ? PB[kk * mB + cj] : 0.0;
                        acc[f] = MFMA(av, bv, acc[f]);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ri = 16 * I + 4 * g + lq, cj = 16 * J + lc;
                        if (ri < m && cj <= m) Tb[(size_t)ri * ld + cj] = acc[f][g];
                    }
                }
            }
            npend = 0;
            __threadfence_block();
            __syncthreads();
            LSTAMP(5);   // fold of KP pending pairs
        }
    }
    LSTAMP(6);
#ifdef QPN_STAMPS
    if (tid == 0 && a.stamps) for (int k = 0; k < 8; ++k) a.stamps[((size_t)a.batch + b) * 8 + k] = stamp_acc[k];
#endif

    // ---- read-back: lambda_k = value of d_k (id m + k)
    __syncthreads();
    double *valv = PA;                                  // 2m + 1 values by id (the pending pairs are done with)
#pragma unroll
    for (int h = 0; h < 2; ++h) if (act[h] && rowvar[h] >= 0) valv[rowvar[h]] = xb[h];
    for (int j = tid; j <= m; j += TPB) valv[colvar[j]] = cnb[j];
    __syncthreads();
    for (int k = tid; k < m; k += TPB) w.lam[vo + k] = valv[m + k];
    if (tid == 0) { w.st2[b] = status; w.piv2[b] = pivots; }
}


// ---- Stage B by BLOCK PRINCIPAL PIVOTING (symmetric Schur problems) ----------------------------------------------------------
// The m x m problem of the kernel above -- s = S lambda + c in [l, u], lambda >= 0 only where s = l, <= 0 only where s = u -- with
// S = Ad H^-1 Ad' SYMMETRIC positive definite on the active rows (the caller vouches for a symmetric H: resident records whose Qd
// blocks are bitwise symmetric).  Lemke's method changes ONE basic variable per pivot and each pivot walks a row and a column
// of the 256 x 257 dictionary (plus its share of the folds): ~100 pivots x 75 KB per node.  Block principal pivoting (Judice &
// Pires) changes ALL violated complementarity pairs at once: with the set A of rows held at a bound,
//       lambda_A = S_AA^-1 (t_A - c_A),   lambda_I = 0,   s = S(:, A) lambda_A + c,
// rows of A whose multiplier has the wrong sign leave, rows outside A that violate their interval enter, and that is repeated
// until nothing changes: 6 - 9 rounds on config 5's nodes, each round a Cholesky factorisation of the |A| <= 112 active rows in
// LDS (16 x 16 tiles, packed lower triangle, the updates on the matrix cores) and two passes over the |A| active ROWS of S
// (coalesced; symmetry turns the column block S(:, A) into rows).  Safeguard: the number of violated pairs has to reach a new
// minimum within three rounds, otherwise only the violated pair with the highest index is changed (Murty's rule: finite for
// the P-matrices at hand).  More than KMAX rows at a bound are admitted lowest index first (the rest waits: the set shrinks
// towards the ~80 that are active at the solution).  A node this kernel does not finish -- a pivot of the factorisation below
// 1e-13 of its diagonal, more than BPP_MAX_IT rounds, m > 256 -- keeps st2 = BPP_NOT_SOLVED and the Lemke kernel takes it as
// before; the post-check on the original blocks certifies the result either way.
constexpr int BPP_KMAX = 112, BPP_TLD = TC_TLD, BPP_TSZ = TC_TSZ, BPP_MAX_IT = 30;
constexpr int BPP_NOT_SOLVED = -7;
__host__ __device__ constexpr int bpp_tiles(int T) { return T * (T + 1) / 2; }
__device__ __forceinline__ int bpp_toff(int i, int j) { return (i * (i + 1) / 2 + j) * BPP_TSZ; }

struct BppShared {
    double rhs[BPP_KMAX + 16];      // t_A - c_A, then y, then lambda_A (by position in the active list)
    int aidx[BPP_KMAX + 16];        // active list: position -> row
    int posof[TPB];                 // row -> position in the active list, or -1
    int cnt[TPB / 64];
    double red[TPB / 64];
    int redi[TPB / 64];
    int fail;
};

// exclusive prefix count of `flag` over the block in thread order, and the total
__device__ __forceinline__ int bpp_prefix(bool flag, int &total, BppShared &B, int tid)
{
    const unsigned long long bal = qpn_ballot(flag);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) B.cnt[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < TPB / 64; ++k) { const int c = B.cnt[k]; if (k < wv) before += c; tot += c; }
    __syncthreads();
    total = tot;
    return before + __popcll(bal & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(TPB, 2) void schur_big_bpp(AviBatchArgs a, SchurBigWs w, double *dict)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    if (a.status[b] != -2) return;
    const int wave = tid >> 6, lane = tid & 63;
    const int m = w.nred[b];
    if (m < 1 || m > TPB || w.s_rowmajor != 2) { if (tid == 0) w.st2[b] = BPP_NOT_SOLVED; return; }
    const int ldp = (m + 1 + 15) & ~15;
    const int ld = ((long long)m * ldp <= (long long)a.N * (a.N + 1)) ? ldp : m + 1;      // (as the Lemke kernel reads T_base)
    const size_t vo = (size_t)b * (size_t)a.N;
    const double *Tb = dict + (size_t)b * (size_t)a.N * (size_t)(a.N + 1);
    __shared__ BppShared B;
    extern __shared__ __attribute__((aligned(16))) double tiles[];       // packed lower triangle of 16 x 16 tiles, row stride 17

    const bool act = tid < m;
    const double ci = act ? w.c[vo + tid] : 0.0;
    const double li = act ? w.l2[vo + tid] : -QINF, ui = act ? w.u2[vo + tid] : QINF;
    // an equality row (its multiplier is free) is not this method's case
    if (__syncthreads_or(act && li == ui)) { if (tid == 0) w.st2[b] = BPP_NOT_SOLVED; return; }
    for (int e = tid; e < bpp_tiles(BPP_KMAX / 16) * BPP_TSZ; e += TPB) tiles[e] = 0.0;
    if (tid == 0) B.fail = 0;
    // scale of the problem for the two tests below
    double cs = act ? fabs(ci) : 0.0;
    if (act && li > -QINF) cs = fmax(cs, fabs(li));
    if (act && ui < QINF) cs = fmax(cs, fabs(ui));
    const double scale = fmax(1.0, sb_block_max(cs, B, tid));
    const double tol_s = 1e-10 * scale;

    int st = 0;                         // 0: lambda = 0, s free in [l, u]; 1: s = l; 2: s = u
    double lam = 0.0, s = ci;
    {
        // the start: the rows violated at lambda = 0 -- when there are more than the cap, the most violated ones (a threshold found
        // by bisection on the count: the prototype needs 5.9 rounds from there, 6.7 - 7.2 from the first 112 by index)
        const double v0 = act ? fmax(li - s, s - ui) : 0.0;
        bool viol = act && v0 > tol_s;
        int total = sb_block_sum_i(viol ? 1 : 0, B, tid);
        if (total > BPP_KMAX) {
            double lo_ = tol_s, hi_ = sb_block_max(v0, B, tid);
            for (int bi = 0; bi < 12; ++bi) {
                const double mid = 0.5 * (lo_ + hi_);
                if (sb_block_sum_i((act && v0 > mid) ? 1 : 0, B, tid) > BPP_KMAX) lo_ = mid; else hi_ = mid;
            }
            viol = act && v0 > hi_;
        }
        if (viol) st = s < li ? 1 : 2;
    }
    int nbest = m + 1, patience = 3, changes = 0;
    bool solved = false, failed = false;
    for (int it = 0; it < BPP_MAX_IT; ++it) {
        // ---- the active list (ascending rows)
        int k;
        const int mypos = bpp_prefix(st != 0, k, B, tid);
        B.posof[tid] = st != 0 ? mypos : -1;
        if (st != 0) { B.aidx[mypos] = tid; B.rhs[mypos] = (st == 1 ? li : ui) - ci; }
        const int T = (k + 15) >> 4, kp = 16 * T;
        if (tid >= k && tid < kp) { B.aidx[tid] = 0; B.rhs[tid] = 0.0; }
        __syncthreads();
        if (k > 0) {
            // ---- S_AA (lower triangle) into the tiles: wave <-> active row, lanes <-> the row's 256 entries (coalesced)
            // (eight rows of a wave in flight: 32 loads, then their scatter)
            for (int a0 = wave; a0 < kp; a0 += 32) {
                double v[8][TPB / 64];
                int pc[TPB / 64];
#pragma unroll
                for (int q = 0; q < TPB / 64; ++q) { const int col = lane + 64 * q; pc[q] = col < m ? B.posof[col] : -1; }
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int ar = a0 + 4 * rr;
                    const double *row = Tb + (size_t)B.aidx[ar < k ? ar : 0] * ld;
#pragma unroll
                    for (int q = 0; q < TPB / 64; ++q) { const int col = lane + 64 * q; v[rr][q] = (ar < k && pc[q] >= 0 && pc[q] <= ar) ? row[col] : 0.0; }
                }
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int ar = a0 + 4 * rr;
                    if (ar >= kp) break;                                  // (wave-uniform)
                    const int ti = ar >> 4, ri = ar & 15;
                    if (ar < k) {
#pragma unroll
                        for (int q = 0; q < TPB / 64; ++q) {
                            const int p = pc[q];
                            if (p >= 0 && p <= ar) tiles[bpp_toff(ti, p >> 4) + ri * BPP_TLD + (p & 15)] = v[rr][q];
                        }
                    } else {
                        // padding up to a whole tile: the identity
                        for (int p = lane; p <= ar; p += 64) tiles[bpp_toff(ti, p >> 4) + ri * BPP_TLD + (p & 15)] = p == ar ? 1.0 : 0.0;
                    }
                }
            }
            __syncthreads();
            // ---- blocked Cholesky and the two triangular solves (qpn_tile_chol.h)
            if (!tc_factor(tiles, T, &B.fail, 1e-13, tid)) { failed = true; break; }
            if (wave == 0) tc_solve(tiles, B.rhs, T, lane);
            __syncthreads();
        }
        // ---- lambda, s = c + S(:, A) lambda_A (S symmetric: rows A of T_base, coalesced over this thread's column)
        lam = st != 0 ? B.rhs[mypos] : 0.0;
        double acc0 = ci, acc1 = 0.0;
        if (act) {
            int a0 = 0;
            for (; a0 + 16 <= k; a0 += 16) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = Tb[(size_t)B.aidx[a0 + q] * ld + tid];
#pragma unroll
                for (int q = 0; q < 16; q += 2) { acc0 = fma(v[q], B.rhs[a0 + q], acc0); acc1 = fma(v[q + 1], B.rhs[a0 + q + 1], acc1); }
            }
            for (; a0 + 4 <= k; a0 += 4) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = Tb[(size_t)B.aidx[a0 + q] * ld + tid];
#pragma unroll
                for (int q = 0; q < 4; q += 2) { acc0 = fma(v[q], B.rhs[a0 + q], acc0); acc1 = fma(v[q + 1], B.rhs[a0 + q + 1], acc1); }
            }
            for (; a0 < k; ++a0) acc0 = fma(Tb[(size_t)B.aidx[a0] * ld + tid], B.rhs[a0], acc0);
        }
        s = st == 1 ? li : (st == 2 ? ui : acc0 + acc1);
        // ---- violated pairs
        const double lmax = sb_block_max(fabs(lam), B, tid);
        const double tol_l = 1e-10 * fmax(1.0, lmax);
        const bool bad_l = (st == 1 && lam < -tol_l) || (st == 2 && lam > tol_l);
        const bool bad_s = act && st == 0 && (s < li - tol_s || s > ui + tol_s);
        const bool bad = bad_l || bad_s;
        const int nb = sb_block_sum_i(bad ? 1 : 0, B, tid);
        if (nb == 0) { solved = true; break; }
        bool flip = bad;
        if (nb < nbest) { nbest = nb; patience = 3; }
        else if (patience > 0) patience--;
        else {
            const int last = -sb_block_min_i(bad ? -tid : 1, B, tid);     // the violated pair with the highest index alone
            flip = bad && tid == last;
        }
        const bool leave = flip && st != 0, enter = flip && st == 0;
        const int nleave = sb_block_sum_i(leave ? 1 : 0, B, tid);
        int nenter;
        const int rank = bpp_prefix(enter, nenter, B, tid);
        const int room = BPP_KMAX - (k - nleave);
        if (leave) st = 0;
        if (enter && rank < room) st = s < li ? 1 : 2;
        changes += nleave + (nenter < room ? nenter : room);
        __syncthreads();                // (B.rhs / aidx are rewritten at the top of the next round)
    }
    if (solved && !failed) {
        if (act) w.lam[vo + tid] = lam;
        if (tid == 0) { w.st2[b] = QPN_SUCCESS; w.piv2[b] = changes; }
    } else if (tid == 0) w.st2[b] = BPP_NOT_SOLVED;
}

} // namespace

size_t qpn_schur_big_workspace_bytes(int batch, int N)
{
    const size_t rows = (size_t)((N + 15) & ~15);
    const size_t tt = (rows < 512 ? rows : 512) * (size_t)(((N + 15) & ~15) + 48);
    return (size_t)batch * (tt + (size_t)N * N + 4 * (size_t)N) * sizeof(double) + (size_t)batch * 4 * sizeof(int32_t) + 1024;
}

// Carves `ws` (qpn_schur_big_workspace_bytes) and runs stage A; fills `out` for the later launches.
hipError_t qpn_launch_schur_big_stage_a(const AviBatchArgs &a, void *ws, SchurBigWs *out, int s_rowmajor, hipStream_t stream)
{
    const int N = a.N, batch = a.batch;
    const size_t rows = (size_t)((N + 15) & ~15);
    SchurBigWs w{};
    w.tt_stride = (int64_t)((rows < 512 ? rows : 512) * (size_t)(((N + 15) & ~15) + 48));
    w.s_stride = (int64_t)N * N;
    w.s_rowmajor = s_rowmajor ? 1 : 0;
    double *p = static_cast<double *>(ws);
    w.Tt = p; p += (size_t)batch * w.tt_stride;
    w.S = p; p += (size_t)batch * w.s_stride;
    w.c = p; p += (size_t)batch * N;
    w.l2 = p; p += (size_t)batch * N;
    w.u2 = p; p += (size_t)batch * N;
    w.lam = p; p += (size_t)batch * N;
    int32_t *ip = reinterpret_cast<int32_t *>(p);
    w.st2 = ip; ip += batch; w.piv2 = ip; ip += batch; w.nsplit = ip; ip += batch; w.nred = ip; ip += batch;
    *out = w;
    static QpnPerDeviceOnce attr_once;
    const int attr_dev = attr_once.device();
    if (!attr_once.done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big_stage_a<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big_stage_a<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big_finish),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024);
        if (e != hipSuccess) return e;
        attr_once.done[attr_dev] = true;
    }
    // node path: every item has the record's (n, m), known here -- small nodes keep their top half in LDS
    if (a.nd.Qd && a.nd.n >= 1 && a.nd.m >= 1 && a.nd.n + a.nd.m == N) {
        const size_t n_pad = (size_t)((a.nd.n + 15) & ~15), m_pad = (size_t)((a.nd.m + 15) & ~15);
        const size_t ldc = n_pad + m_pad + 16;
        const size_t up = 2 * n_pad * LDU + ((2 * n_pad * LDU) & 1);          // two U' panels, then an even offset
        const size_t bytes = (up + n_pad * ldc) * sizeof(double);
        if (bytes <= 75 * 1024) {                       // two workgroups per CU (one is no faster than the HBM route)
            hipLaunchKernelGGL(schur_big_stage_a<true>, dim3((unsigned)batch), dim3(TPB), bytes, stream, a, w,
                               (int)(2 * n_pad), (int)up, (int)(n_pad * ldc));
            return hipGetLastError();
        }
    }
    const size_t rows_lds = rows < 512 ? rows : 512;
    size_t lds_dbl = rows_lds * LDU;
    if (lds_dbl < 4 * 1056) lds_dbl = 4 * 1056;          // the fill's four transpose tiles
    hipLaunchKernelGGL(schur_big_stage_a<false>, dim3((unsigned)batch), dim3(TPB), lds_dbl * sizeof(double), stream, a, w,
                       (int)rows_lds, 0, 0);
    return hipGetLastError();
}

hipError_t qpn_launch_schur_big_finish(const AviBatchArgs &a, const SchurBigWs &w, hipStream_t stream)
{
    hipLaunchKernelGGL(schur_big_finish, dim3((unsigned)a.batch), dim3(TPB), (size_t)a.N * sizeof(double), stream, a, w);
    return hipGetLastError();
}

// Lemke on the Schur problems of the accepted items (status -2) with delayed updates; `dict` is the large-item
// kernel's dictionary workspace (batch x N x (N+1) doubles), reused here as T_base.
hipError_t qpn_launch_schur_big_lemke(const AviBatchArgs &a, const SchurBigWs &w, double *dict, hipStream_t stream, int after_bpp)
{
    static QpnPerDeviceOnce attr_once;
    const int attr_dev = attr_once.device();
    if (!attr_once.done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big_lemke<16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big_lemke<8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_once.done[attr_dev] = true;
    }
    // The reduced size m of an item is known on the device only (stage A found the split), and the LDS of a launch
    // is sized on the host: two launches, each taking the items of its own m range -- m <= ceil(N / 2) (a node
    // with no more constraint rows than variables: config 5) gets 16 pending pairs in <= 80 KB, i.e. two
    // workgroups per CU, the rest is sized for the largest m the path takes.  A launch with no item is a few us.
    const int N = a.N;
    const int m_top = N - 1 < 512 ? N - 1 : 512;         // stage A declines larger ones
    const int m_mid = (N + 1) / 2 < m_top ? (N + 1) / 2 : m_top;
    auto bytes_for = [](int KP, int mh) -> size_t {
        const size_t mB = (size_t)mh + 1;
        const size_t dbl = (size_t)KP * 2 * (size_t)sb_pend_stride(mh) + (mB + 2) + 2 * mB;
        return ((dbl * sizeof(double) + sizeof(int) * (2 * mB + 4)) + 15) & ~(size_t)15;
    };
    for (int part = 0; part < 2; ++part) {
        const int m_lo = part == 0 ? 1 : m_mid + 1, m_hi = part == 0 ? m_mid : m_top;
        if (m_lo > m_hi) continue;
        const int KP = bytes_for(16, m_hi) <= 78 * 1024 ? 16 : 8;
        const size_t bytes = bytes_for(KP, m_hi);
        if (bytes > 150 * 1024) return hipErrorInvalidValue;
        if (KP == 16) hipLaunchKernelGGL(schur_big_lemke<16>, dim3((unsigned)a.batch), dim3(TPB), bytes, stream, a, w, dict, m_lo, m_hi, after_bpp);
        else hipLaunchKernelGGL(schur_big_lemke<8>, dim3((unsigned)a.batch), dim3(TPB), bytes, stream, a, w, dict, m_lo, m_hi, after_bpp);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Stage B by block principal pivoting for the nodes whose Schur problem is symmetric (the caller's promise); what it does not
// finish keeps st2 = BPP_NOT_SOLVED for the Lemke kernel (launch it with after_bpp = 1).
hipError_t qpn_launch_schur_big_bpp(const AviBatchArgs &a, const SchurBigWs &w, double *dict, hipStream_t stream)
{
    const size_t bytes = (size_t)bpp_tiles(BPP_KMAX / 16) * BPP_TSZ * sizeof(double);       // 60 928 B: two workgroups per CU
    static QpnPerDeviceOnce attr_once;
    const int attr_dev = attr_once.device();
    if (!attr_once.done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big_bpp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_once.done[attr_dev] = true;
    }
    hipLaunchKernelGGL(schur_big_bpp, dim3((unsigned)a.batch), dim3(TPB), bytes, stream, a, w, dict);
    return hipGetLastError();
}
