// qpn_avi_schur_mid.hip -- the crash of MID-SIZE nodes (n, m <= 64, at least one of them > 32) on the matrix cores, gfx950.
//
// qpn_solve_nodes for node records [free x_d x n | GAVI x m] that do not fit the one-wavefront kernel of
// qpn_avi_schur.hip (n, m <= 32) used to take the route built for config 5 (qpn_avi_schur_big.hip): assemble M in
// HBM, a workgroup per item that walks 16-wide panels of a row-major top half through LDS with ~10 barriers per
// panel, the Schur problem to the register kernel, a finish kernel on the assembled blocks -- 3.8 M solves/s at
// n = m = 48 against 90 M/s at n = m = 32.  Here the same algebra is laid out for THIS size class:
//   stage_a  one workgroup = 4 wavefronts per node, straight from the node records (no assembled M):
//     * wave v owns ROW TILE v of the top half [H | C~ | g] (16 rows x up to 128 columns = 8 MFMA tiles = 64 VGPRs),
//       C~ = +Ad' so that the eliminated tiles hold W~ = -W = H^-1 Ad' (as in the node path of qpn_avi_schur.hip);
//     * rank-4 block pivots exactly as in the 32-class kernel (4 x 4 LU without pivoting, every pivot must pass
//       |u_ss| >= 1e-4 max(1, max|M|), else the node is declined; pivot rows carry P - I so that the update turns them
//       into P^-1 V): the owner wave publishes the raw pivot rows and the pivot block through LDS (double buffered:
//       ONE workgroup barrier per step, n/4 <= 16 in all), every wave then updates its own row tile with one MFMA per
//       live column tile;
//     * S = Ad W~ and c = b - Ad h: W~ goes through LDS once (32 KB at 64 x 64, blocks rotated by the row so that
//       the B-operand reads are conflict-free), wave v computes row tile v of S; S (column-major), c, W~ and h go to
//       a workspace;
//   the Schur problems (all GAVI rows, size m <= 64) run on the one-wavefront register kernel (qpn_avi_reg.hip,
//     same pivot rule), bounds read straight from the records;
//   finish   x = W~ lambda - h, post-check / residual / active sets on the ORIGINAL blocks taken from the records
//     ([[Qd, -Ad'],[Ad, 0]], q = [qd + R w; B w]: src/avi.jl:205-251 + :305-377), same arithmetic order as the
//     general finish.
// Declined nodes (a block pivot below the threshold, an equality row) keep status -1 and go to the general path in
// gated launches (assembly + general kernel), exactly as for the 32-class kernel.
#include "qpn_internal.h"
#include <cstdlib>

#define QINF __builtin_huge_val()

namespace {

constexpr int TPB = 256;
constexpr int VLD = 144;                // row stride of the published pivot rows (128 columns; == 16 mod 32: the two
                                        // row groups of a half-wave read disjoint banks)
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)
#define MFMA_NEGA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 1)      // D = C - A B (gfx950 NEG bits)

__device__ __forceinline__ int pad16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double rcp64(double x)      // one Newton step on v_rcp_f64: <= 10 ulp (tools/rcp_probe.hip)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

#ifdef QPN_STAMPS
#define STAMP(slot)                                                     \
    do {                                                                \
        unsigned long long now__ = __builtin_amdgcn_s_memtime();        \
        __builtin_amdgcn_s_waitcnt(0xC07F);                             \
        stamp_acc[slot] += now__ - stamp_last;                          \
        stamp_last = now__;                                             \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// LDS map (doubles).  sW (the eliminated W~, for the S product) reuses the area of the published pivot rows.
constexpr int OFF_Q = 0;                // q = [g ; b] in item order                       [128]
constexpr int OFF_U = 128;              // per wave: its 16 x 4 panel of pivot columns     [4][64]
constexpr int OFF_P = 384;              // inverse of the pivot block, two buffers         [2][16]
constexpr int OFF_X = 416;              // P^-1 x_piv (extra-column entries of the pivot rows) [2][4]
constexpr int OFF_RED = 424;            // block reduction [4], pivot-test flags [2]       [8]
constexpr int OFF_H = 432;              // h (the eliminated extra column)                 [64]
constexpr int OFF_PR = 496;             // raw pivot block + x_piv (owner wave only)       [20]
constexpr int OFF_BIG = 516;            // Qd staging [n][n_pad + 2]  /  pivot rows [2][4][VLD]  /  W~ [n_pad][m_pad]
__host__ __device__ constexpr int mid_lds_doubles(int n, int n_pad, int m_pad)
{
    int big = 2 * 4 * VLD;
    if (n_pad * m_pad > big) big = n_pad * m_pad;
    if (n_pad * (n_pad + 2) > big) big = n_pad * (n_pad + 2);
    return OFF_BIG + big;
}

// TM: upper bound of the tile counts of a launch (n_pad / 16 and m_pad / 16 <= TM): the 48-class (TM = 3) carries neither the
// registers nor the guards of the fourth tiles
template <int TM, bool BLK>
__global__ __launch_bounds__(TPB, 4) void schur_mid_stage_a(AviBatchArgs a, SchurMidWs w)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    const int v = tid >> 6, l = tid & 63, lc = l & 15, lq = l >> 4;
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    const int n_pad = pad16(n), m_pad = pad16(m), nht = n_pad >> 4, mct = m_pad >> 4;
    extern __shared__ __attribute__((aligned(32))) double sm[];
    double *const sQ = sm + OFF_Q, *const sUv = sm + OFF_U + 64 * v, *const sP = sm + OFF_P, *const sX = sm + OFF_X;
    double *const sRed = sm + OFF_RED, *const sFlag = sm + OFF_RED + 4, *const sPr = sm + OFF_PR, *const sH = sm + OFF_H, *const sV = sm + OFF_BIG, *const sW = sm + OFF_BIG;

    const double *Q_ = a.nd.Qd + (size_t)b * n * n;
    const double *A_ = a.nd.Ad + (size_t)b * m * n;
    const double *R_ = a.nd.R + (size_t)b * n * np_;
    const double *B_ = a.nd.B + (size_t)b * m * np_;
    const double *w_ = a.nd.w + (size_t)b * (size_t)a.nd.stride_w;
    auto decline = [&]() {
        if (tid == 0) {
            a.status[b] = -1;
            if (a.decl_count) atomicAdd(a.decl_count, 1);
        }
    };
    if (b == 0 && tid < 64) w.ones[tid] = 1;               // kind vector of the reduced problems (all GAVI)
#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif

    // ---- load: wave v takes row tile v of [H | C~] straight into the MFMA C/D layout --------------------------
    // H(r, c) = Qd[c * n + r] (padded rows: identity), C~(r, k) = Ad[r * m + k] (Ad is m x n column-major)
    d4 th0, th1, th2, th3, tc0, tc1, tc2, tc3;
    double mabs = 0.0;
    // C~ tiles: direct loads (a row group of a tile is 128 contiguous bytes of Ad)
#define M_LOADC(J, T)                                                                               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * v + 4 * g + lq, ck = 16 * (J) + lc;                                     \
        const bool valid = (J) < TM && rr < n && ck < m;                                            \
        double t_ = 0.0;                                                                            \
        if ((J) < TM) t_ = A_[valid ? (size_t)rr * m + ck : 0];                                     \
        T[g] = valid ? t_ : 0.0;                                                                    \
    }
    M_LOADC(0, tc0) M_LOADC(1, tc1) M_LOADC(2, tc2) M_LOADC(3, tc3)
#undef M_LOADC
    // Qd: whole columns with coalesced loads (lane <-> row, the four waves take every fourth column, all loads of a
    // thread in flight together) into LDS, column stride n_pad + 2 (conflict-free tile reads); tiles from there
    const int LDQ = n_pad + 2;
    double *const sQd = sm + OFF_BIG;
    {
        // the padded n_pad x n_pad block: entries outside Qd are the identity (padded rows pivot on themselves)
        double vq[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = v + 4 * t;
            const bool ok = t < 4 * TM && j < n && l < n;
            double t_ = 0.0;
            if (t < 4 * TM) t_ = Q_[ok ? (size_t)j * n + l : 0];
            vq[t] = ok ? t_ : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = v + 4 * t;
            mabs = fmax(mabs, fabs(vq[t]));
            if (j == l && l >= n) vq[t] = 1.0;
            if (t < 4 * TM && j < n_pad && l < n_pad) sQd[j * LDQ + l] = vq[t];
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        mabs = fmax(fmax(mabs, fabs(tc0[g])), fmax(fmax(fabs(tc1[g]), fabs(tc2[g])), fabs(tc3[g])));
    __syncthreads();
    const d4 z4 = {0.0, 0.0, 0.0, 0.0};
#define M_LOADH(J, T)                                                                               \
    T = z4;                                                                                         \
    if ((J) < TM && v < nht && (J) < nht) {                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) T[g] = sQd[(16 * (J) + lc) * LDQ + 16 * v + 4 * g + lq]; \
    }
    M_LOADH(0, th0) M_LOADH(1, th1) M_LOADH(2, th2) M_LOADH(3, th3)
#undef M_LOADH
    // q = [qd + R w; B w], the p terms in ascending order (the fma chain of the assembly kernel), eight loads in flight
    double qkeep = 0.0;                 // stored at the very end: a barrier waits for every outstanding global store
    if (tid < N) {
        const bool isx = tid < n;
        const double *col = isx ? R_ + tid : B_ + (tid - n);
        const size_t cs = isx ? (size_t)n : (size_t)m;
        double s = isx ? a.nd.qd[(size_t)b * n + tid] : 0.0;
        for (int k0 = 0; k0 < np_; k0 += 8) {
            double rv[8], wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool ok = k0 + k < np_;
                rv[k] = col[ok ? (size_t)(k0 + k) * cs : 0];
                wv[k] = w_[ok ? k0 + k : 0];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) s = (k0 + k < np_) ? fma(rv[k], wv[k], s) : s;
        }
        sQ[tid] = s;
        qkeep = s;
    }
    // equality GAVI rows need their multiplier crashed in: left to the general kernel
    bool eqrow = false;
    if (tid < m) eqrow = a.nd.l[(size_t)b * m + tid] == a.nd.u[(size_t)b * m + tid];
    {
        double r = mabs;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(r, off, 64); r = o > r ? o : r; }
        if (l == 0) sRed[v] = r;
    }
    if (__syncthreads_or(eqrow ? 1 : 0)) { decline(); return; }
    const double mscale = fmax(fmax(sRed[0], sRed[1]), fmax(sRed[2], sRed[3]));
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);
    double kx = (l < 16 && 16 * v + l < n) ? sQ[16 * v + l] : 0.0;       // lane l <-> row 16 v + l of the extra column
    STAMP(0);   // load

    // ---- rank-4 block pivots ---------------------------------------------------------------------------------
    bool fail = false;
#define M_PUB(J, T, GP) if ((J) < TM && (J) < nht) sVp[lq * VLD + 16 * (J) + lc] = T[GP];
#define M_PUBC(J, T, GP) if ((J) < TM && (J) < mct) sVp[lq * VLD + 64 + 16 * (J) + lc] = T[GP];
#define M_UPD(J, T) if ((J) < TM && (J) < nht) { const double vr_ = sVp[lq * VLD + 16 * (J) + lc]; T = MFMA_NEGA(au, vr_, T); }
#define M_UPDC(J, T) if ((J) < TM && (J) < mct) { const double vr_ = sVp[lq * VLD + 64 + 16 * (J) + lc]; T = MFMA_NEGA(au, vr_, T); }
#define M_STEP(KB, THJP)                                                                            \
    if ((KB) / 4 < TM && !fail && 4 * (KB) < n) {                                                   \
        constexpr int JP = (KB) / 4, GP = (KB) % 4, cq = 4 * GP, par = (KB) & 1;                    \
        double *const sVp = sV + par * 4 * VLD;                                                     \
        double *const sPp = sP + par * 16;              /* P^-1 transposed: [column][row] */          \
        double *const sXp = sX + par * 4;               /* P^-1 x_piv */                             \
        if (v < nht) {                                                                              \
            const int kcol = lc - cq;                                                               \
            if (kcol >= 0 && kcol < 4) {                                                            \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                     \
                    double val = THJP[g];                                                           \
                    if (v == JP && g == GP) { sPr[lq * 4 + kcol] = val; if (lq == kcol) val -= 1.0; } \
                    sUv[(4 * g + lq) * 4 + kcol] = val;                                             \
                }                                                                                   \
            }                                                                                       \
            if (v == JP) {          /* the owner of the pivot rows */                               \
                /* their raw values in every live column tile */                                    \
                if (JP <= 0) M_PUB(0, th0, GP)                                                      \
                if (JP <= 1) M_PUB(1, th1, GP)                                                      \
                if (JP <= 2) M_PUB(2, th2, GP)                                                      \
                M_PUB(3, th3, GP)                                                                   \
                M_PUBC(0, tc0, GP) M_PUBC(1, tc1, GP) M_PUBC(2, tc2, GP) M_PUBC(3, tc3, GP)         \
                if (l >= cq && l < cq + 4) sPr[16 + l - cq] = kx;       /* their extra-column entries x_piv */ \
                /* P = L U (unit lower L, no pivoting) with lane <-> row (l & 3), pivot rows broadcast with v_readlane; \
                   then lane c < 4 solves L y = e_c, U x = y (column c of P^-1) and lanes 4..7 the same with x_piv */ \
                wave_sync();                                                                        \
                const int li = l & 3;                                                               \
                double pr[4];                                                                       \
                { const d4 row = *reinterpret_cast<const d4 *>(sPr + li * 4); pr[0] = row[0]; pr[1] = row[1]; pr[2] = row[2]; pr[3] = row[3]; } \
                const d4 xraw = *reinterpret_cast<const d4 *>(sPr + 16);                            \
                bool okp = true;                                                                    \
                double rd[4];                                                                       \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                     \
                    const double piv = readlane_f64(pr[s], s);                                      \
                    okp = okp && fabs(piv) >= diag_thr;                                             \
                    rd[s] = udbl(rcp64(piv));                                                       \
                    const double f = pr[s] * rd[s];                                                 \
                    const bool below = li > s;                                                      \
                    if (below) pr[s] = f;                                                           \
                    _Pragma("unroll") for (int j = s + 1; j < 4; ++j) {                             \
                        const double psj = readlane_f64(pr[j], s);                                  \
                        if (below) pr[j] = fma(-f, psj, pr[j]);                                     \
                    }                                                                               \
                }                                                                                   \
                double y[4];                                                                        \
                _Pragma("unroll") for (int j2 = 0; j2 < 4; ++j2) {                                  \
                    double sacc = (l & 4) ? xraw[j2] : ((j2 == li) ? 1.0 : 0.0);                    \
                    _Pragma("unroll") for (int i2 = 0; i2 < j2; ++i2) sacc = fma(-readlane_f64(pr[i2], j2), y[i2], sacc); \
                    y[j2] = sacc;                                                                   \
                }                                                                                   \
                _Pragma("unroll") for (int j2 = 3; j2 >= 0; --j2) {                                 \
                    double sacc = y[j2];                                                            \
                    _Pragma("unroll") for (int i2 = j2 + 1; i2 < 4; ++i2) sacc = fma(-readlane_f64(pr[i2], j2), y[i2], sacc); \
                    y[j2] = sacc * rd[j2];                                                          \
                }                                                                                   \
                if (l < 5) {                                                                        \
                    d4 yv; yv[0] = y[0]; yv[1] = y[1]; yv[2] = y[2]; yv[3] = y[3];                  \
                    *reinterpret_cast<d4 *>(l < 4 ? sPp + 4 * l : sXp) = yv;                        \
                }                                                                                   \
                if (l == 0) sFlag[par] = okp ? 1.0 : 0.0;                                           \
            }                                                                                       \
        }                                                                                           \
        __syncthreads();                                                                            \
        STAMP(1);   /* gather + factor + publish + barrier */                                       \
        if (sFlag[par] == 0.0) { fail = true; }          /* uniform over the workgroup */            \
        else if (v < nht) {                                                                         \
            /* U' = U P^-1 straight into the A-operand layout: lane (lc, lq) forms U'[row lc][column lq] (pivot rows \
               hold P - I, so theirs is I - P^-1); extra column: kx_l -= U[l] . (P^-1 x_piv) */      \
            const d4 ur = *reinterpret_cast<const d4 *>(sUv + lc * 4);                              \
            const d4 pt = *reinterpret_cast<const d4 *>(sPp + lq * 4);                              \
            const d4 tx = *reinterpret_cast<const d4 *>(sXp);                                       \
            const double au = fma(ur[3], pt[3], fma(ur[2], pt[2], fma(ur[1], pt[1], ur[0] * pt[0]))); \
            kx -= fma(ur[3], tx[3], fma(ur[2], tx[2], fma(ur[1], tx[1], ur[0] * tx[0])));           \
            STAMP(2);   /* U' */                                                                    \
            if (JP <= 0) M_UPD(0, th0)                                                              \
            if (JP <= 1) M_UPD(1, th1)                                                              \
            if (JP <= 2) M_UPD(2, th2)                                                              \
            M_UPD(3, th3)                                                                           \
            M_UPDC(0, tc0) M_UPDC(1, tc1) M_UPDC(2, tc2) M_UPDC(3, tc3)                             \
            wave_sync();                                                                            \
            STAMP(3);   /* tile updates */                                                          \
        }                                                                                           \
    }
    M_STEP(0, th0) M_STEP(1, th0) M_STEP(2, th0) M_STEP(3, th0)
    M_STEP(4, th1) M_STEP(5, th1) M_STEP(6, th1) M_STEP(7, th1)
    M_STEP(8, th2) M_STEP(9, th2) M_STEP(10, th2) M_STEP(11, th2)
    M_STEP(12, th3) M_STEP(13, th3) M_STEP(14, th3) M_STEP(15, th3)
#undef M_STEP
#undef M_PUB
#undef M_PUBC
#undef M_UPD
#undef M_UPDC
    if (fail) { decline(); return; }

    // A operands of the S product (16 x 4 blocks of Ad, element (i = lc, k = lq)): all of them requested now, so that the
    // round trip hides behind the W~ hand-over (the H tiles are dead: their registers are free)
    const int arow = 16 * v + lc;
    double aop[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int r = 4 * kk + lq;                                  // k index: column of Ad, row of W~
        const bool valid = kk < 4 * TM && v < mct && r < n && arow < m;
        const double t_ = A_[valid ? (size_t)r * m + arow : 0];
        aop[kk] = valid ? t_ : 0.0;
    }
    // ---- W~ and h: to LDS for the S product, to the workspace for the finish kernel ----------------------------
    __syncthreads();                                   // the last step's pivot rows have been read (sW reuses them)
    const int rot = mct == 4 ? lq : (mct == 3 ? (lq == 3 ? 0 : lq) : (mct == 2 ? (lq & 1) : 0));     // lq mod mct
    double *const Wg = w.W + (size_t)b * (size_t)w.w_stride;       // W~ column-major [m_pad][n_pad], then h [n_pad]
    if (v < nht) {
#define M_WOUT(J, T)                                                                                \
    if ((J) < TM && (J) < mct) {                                                                    \
        int jj = (J) + rot; if (jj >= mct) jj -= mct;                                               \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
            const int rr = 16 * v + 4 * g + lq;                                                     \
            sW[rr * m_pad + 16 * jj + lc] = T[g];                                                   \
        }                                                                                           \
    }
        M_WOUT(0, tc0) M_WOUT(1, tc1) M_WOUT(2, tc2) M_WOUT(3, tc3)
#undef M_WOUT
        if (l < 16) sH[16 * v + l] = kx;
    }
    __syncthreads();
    if (v < nht) {                  // the workspace copies drain behind the S product (no barrier after them)
#define M_WOUT(J, T)                                                                                \
    if ((J) < TM && (J) < mct) {                                                                    \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                               \
            Wg[(size_t)(16 * (J) + lc) * n_pad + 16 * v + 4 * g + lq] = T[g];                       \
    }
        M_WOUT(0, tc0) M_WOUT(1, tc1) M_WOUT(2, tc2) M_WOUT(3, tc3)
#undef M_WOUT
        if (l < 16) Wg[(size_t)m_pad * n_pad + 16 * v + l] = kx;
    }
    if (tid < N) w.gq[(size_t)b * N + tid] = qkeep;
    STAMP(4);   // W~ out

    // ---- S = Ad W~ (row tile v), c = b - Ad h ------------------------------------------------------------------
    if (v < mct) {
        d4 s0 = z4, s1 = z4, s2 = z4, s3 = z4, sx;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ri = 16 * v + 4 * g + lq;
            sx[g] = (lc == 0 && ri < m) ? sQ[n + ri] : 0.0;
        }
        int j0 = rot, j1 = 1 + rot, j2 = 2 + rot, j3 = 3 + rot;
        if (j0 >= mct) j0 -= mct;
        if (j1 >= mct) j1 -= mct;
        if (j2 >= mct) j2 -= mct;
        if (j3 >= mct) j3 -= mct;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            if (kk < 4 * TM && 4 * kk < n_pad) {
                const int r = 4 * kk + lq;
                const double a_ = aop[kk];
                const double *wr = sW + r * m_pad + lc;
                s0 = MFMA(a_, wr[16 * j0], s0);
                if (mct > 1) s1 = MFMA(a_, wr[16 * j1], s1);
                if (mct > 2) s2 = MFMA(a_, wr[16 * j2], s2);
                if (TM > 3 && mct > 3) s3 = MFMA(a_, wr[16 * j3], s3);
                const double hb = lc == 0 ? sH[r] : 0.0;
                sx = MFMA_NEGA(a_, hb, sx);
            }
        }
        double *Sg = w.S + (size_t)b * 4096;
        // S for the Lemke kernel: its register-block layout [cj % bs][ri % bs][lane = 8 (ri / bs) + cj / bs] (s_bs > 0), else
        // column-major; x / bs as a multiply-shift (x < 64)
        const int sbs = BLK ? w.s_bs : 1;
        const unsigned minv = (65536u + (unsigned)sbs - 1u) / (unsigned)sbs;
#define M_SOUT(J, T)                                                                                \
    if ((J) < TM && (J) < mct) {                                                                    \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
            const int ri = 16 * v + 4 * g + lq, cj = 16 * (J) + lc;                                 \
            int ib_ = cj * m + ri;                                                                  \
            if constexpr (BLK) {                                                                    \
                const int rb_ = (int)(((unsigned)ri * minv) >> 16), cb_ = (int)(((unsigned)cj * minv) >> 16); \
                ib_ = ((cj - cb_ * sbs) * sbs + (ri - rb_ * sbs)) * 64 + rb_ * 8 + cb_;            \
            }                                                                                       \
            if (ri < m && cj < m) Sg[ib_] = T[g];                                                   \
        }                                                                                           \
    }
        M_SOUT(0, s0) M_SOUT(1, s1) M_SOUT(2, s2) M_SOUT(3, s3)
#undef M_SOUT
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ri = 16 * v + 4 * g + lq;
            if (lc == 0 && ri < m) w.c[(size_t)b * m + ri] = sx[g];
        }
    }
    STAMP(5);   // S product + stores
#ifdef QPN_STAMPS
    if (tid == 0 && a.stamps) {
        for (int k = 0; k < 8; ++k) a.stamps[(size_t)b * 8 + k] = stamp_acc[k];
    }
#endif
    if (tid == 0) a.status[b] = -2;
}

// ---- finish: x = W~ lambda - h, post-check on the original blocks (from the records) ----------------------------
constexpr int TPF = 128;
__global__ __launch_bounds__(TPF) void schur_mid_finish(AviBatchArgs a, SchurMidWs w)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    if (a.status[b] != -2) return;
    const int n = a.nd.n, m = a.nd.m, N = n + m;
    const int n_pad = pad16(n), m_pad = pad16(m);
    const int LDA = m | 1;
    extern __shared__ __attribute__((aligned(16))) double fs[];
    double *const zs = fs;                    // z, N doubles (<= 128)
    double *const red = fs + 128;             // [4]
    double *const sAd = fs + 136;             // Ad, column stride LDA (odd: conflict-free along rows AND along columns)
    const double *Q_ = a.nd.Qd + (size_t)b * n * n;
    const double *A_ = a.nd.Ad + (size_t)b * m * n;
    const double *Wg = w.W + (size_t)b * (size_t)w.w_stride;
    for (int e0 = tid; e0 < m * n; e0 += 4 * TPF) {
        double vv[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) { const int e = e0 + q4 * TPF; vv[q4] = e < m * n ? A_[e] : 0.0; }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int e = e0 + q4 * TPF;
            const int j = e / m, i = e - j * m;
            if (e < m * n) sAd[j * LDA + i] = vv[q4];
        }
    }
    if (tid < m) zs[n + tid] = w.lam[(size_t)b * m + tid];
    __syncthreads();
    if (tid < n) {
        double s = -Wg[(size_t)m_pad * n_pad + tid];
        int k = 0;
        for (; k + 8 <= m; k += 8) {
            double wv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) wv[q8] = Wg[(size_t)(k + q8) * n_pad + tid];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) s = fma(wv[q8], zs[n + k + q8], s);
        }
        for (; k + 4 <= m; k += 4) {
            double wv[4];
#pragma unroll
            for (int q8 = 0; q8 < 4; ++q8) wv[q8] = Wg[(size_t)(k + q8) * n_pad + tid];
#pragma unroll
            for (int q8 = 0; q8 < 4; ++q8) s = fma(wv[q8], zs[n + k + q8], s);
        }
        for (; k < m; ++k) s = fma(Wg[(size_t)k * n_pad + tid], zs[n + k], s);
        zs[tid] = s;
    }
    __syncthreads();
    int bad = 0;
    double nres = 0.0;
    if (tid < N) {
        const int k = tid;
        const int gk = k >= n;
        double rk = w.gq[(size_t)b * N + k];
        // row k of [[Qd, -Ad'],[Ad, 0]] times z, columns ascending; a zero z_j contributes nothing (as in the general finish)
        if (!gk) {
            int j = 0;
            for (; j + 8 <= n; j += 8) {
                double mv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) mv[q8] = Q_[(size_t)(j + q8) * n + k];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) { const double zj = zs[j + q8]; rk = (zj != 0.0) ? fma(mv[q8], zj, rk) : rk; }
            }
            for (; j + 4 <= n; j += 4) {
                double mv[4];
#pragma unroll
                for (int q8 = 0; q8 < 4; ++q8) mv[q8] = Q_[(size_t)(j + q8) * n + k];
#pragma unroll
                for (int q8 = 0; q8 < 4; ++q8) { const double zj = zs[j + q8]; rk = (zj != 0.0) ? fma(mv[q8], zj, rk) : rk; }
            }
            for (; j < n; ++j) { const double zj = zs[j]; if (zj != 0.0) rk = fma(Q_[(size_t)j * n + k], zj, rk); }
            for (int i = 0; i < m; ++i) { const double zj = zs[n + i]; if (zj != 0.0) rk = fma(-sAd[k * LDA + i], zj, rk); }
        } else {
            const int r = k - n;
            for (int j = 0; j < n; ++j) { const double zj = zs[j]; if (zj != 0.0) rk = fma(sAd[j * LDA + r], zj, rk); }
        }
        const double zk = zs[k];
        const double lk = gk ? a.nd.l[(size_t)b * m + (k - n)] : -QINF, uk = gk ? a.nd.u[(size_t)b * m + (k - n)] : QINF;
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
        nres = e;
        unsigned mask = 0;
        const double ct = a.comp_tol;
        auto approx = [&](double x, double y) { return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= ct); };
        if (!approx(lk, uk)) {
            if (approx(p, lk) && d >= -ct) mask |= 1u;
            if (lk - ct <= p && p <= uk + ct && fabs(d) <= ct) mask |= 2u;
            if (approx(p, uk) && d <= ct) mask |= 4u;
        } else mask = 8u;
        if (gk) mask <<= 4;
        a.z[(size_t)b * N + k] = zk;
        if (a.active) a.active[(size_t)b * N + k] = (uint8_t)mask;
    }
    const int badt = __syncthreads_count(bad > 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(nres, off, 64); nres = o > nres ? o : nres; }
    if ((tid & 63) == 0) red[tid >> 6] = nres;
    __syncthreads();
    if (tid == 0) {
        int status = w.st2[b];
        if (badt > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
        a.status[b] = status;
        if (a.resid) a.resid[b] = red[0] > red[1] ? red[0] : red[1];
        if (a.pivots) a.pivots[b] = n + w.piv2[b];
    }
}

} // namespace

bool qpn_schur_mid_shape(int n, int m)
{
    return (n > 32 || m > 32) && n >= 1 && n <= 64 && m >= 1 && m <= 64;
}

size_t qpn_schur_mid_workspace_bytes(int batch, int n, int m)
{
    const size_t n_pad = (size_t)((n + 15) & ~15), m_pad = (size_t)((m + 15) & ~15);
    const size_t per = n_pad * m_pad + n_pad + 4096 + 2 * (size_t)m + (size_t)(n + m);
    return (size_t)batch * per * sizeof(double) + (size_t)batch * 2 * sizeof(int32_t) + 512;
}

// Stage A of every node, the Schur problems on the register kernel, finish.  Nodes it declines keep status -1.
hipError_t qpn_launch_schur_mid_nodes(const AviBatchArgs &a, void *ws, hipStream_t stream)
{
    const int n = a.nd.n, m = a.nd.m, batch = a.batch;
    if (batch <= 0) return hipSuccess;
    const size_t n_pad = (size_t)((n + 15) & ~15), m_pad = (size_t)((m + 15) & ~15);
    SchurMidWs w{};
    w.w_stride = (int64_t)(n_pad * m_pad + n_pad);
    double *p = static_cast<double *>(ws);
    w.W = p; p += (size_t)batch * w.w_stride;
    w.S = p; p += (size_t)batch * 4096;
    w.s_bs = qpn_avi_reg_block_size(m) < 8 ? qpn_avi_reg_block_size(m) : 0;     // (the 8 x 8 class has its own full-width staged load)
    w.c = p; p += (size_t)batch * m;
    w.lam = p; p += (size_t)batch * m;
    w.gq = p; p += (size_t)batch * (n + m);
    int32_t *ip = reinterpret_cast<int32_t *>(p);
    w.st2 = ip; ip += batch; w.piv2 = ip; ip += batch;
    w.ones = reinterpret_cast<uint8_t *>(ip);
    const size_t lds_a = (size_t)mid_lds_doubles(n, (int)n_pad, (int)m_pad) * sizeof(double);
    const size_t lds_f = (136 + (size_t)n * (size_t)(m | 1)) * sizeof(double);
    const bool t3 = n_pad <= 48 && m_pad <= 48, blk = w.s_bs > 0;
    if (t3 && blk) hipLaunchKernelGGL((schur_mid_stage_a<3, true>), dim3((unsigned)batch), dim3(TPB), lds_a, stream, a, w);
    else if (t3) hipLaunchKernelGGL((schur_mid_stage_a<3, false>), dim3((unsigned)batch), dim3(TPB), lds_a, stream, a, w);
    else if (blk) hipLaunchKernelGGL((schur_mid_stage_a<4, true>), dim3((unsigned)batch), dim3(TPB), lds_a, stream, a, w);
    else hipLaunchKernelGGL((schur_mid_stage_a<4, false>), dim3((unsigned)batch), dim3(TPB), lds_a, stream, a, w);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    AviBatchArgs r{};
    r.batch = batch; r.N = m; r.vec_stride = m;
    r.M = w.S; r.strideM = 4096; r.q = w.c; r.l = a.nd.l; r.u = a.nd.u; r.kind = w.ones; r.stride_kind = 0;
    r.z = w.lam; r.status = w.st2; r.pivots = w.piv2; r.resid = nullptr; r.active = nullptr;
    r.check_tol = a.check_tol; r.piv_tol = a.piv_tol; r.feas_tol = a.feas_tol; r.comp_tol = a.comp_tol;
    r.max_pivots = (a.max_pivots > 0 ? a.max_pivots : 50 * (n + m) + 100) - n;        // the crash pivots count
    r.flags = a.flags | QPN_AVI_FLAG_COLD_START | (w.s_bs > 0 ? QPN_AVI_IFLAG_BLOCKED_M : 0);
    r.only_if = a.status; r.only_if_value = -2;
#ifdef QPN_STAMPS
    if (QPN_DEV_ENV("QPN_MID_STAMP_REG")) r.stamps = a.stamps;          // diagnostic builds: the Lemke kernel's phases instead
#endif
    e = qpn_launch_avi_solve_reg(r, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(schur_mid_finish, dim3((unsigned)batch), dim3(TPF), lds_f, stream, a, w);
    return hipGetLastError();
}
