// qpn_avi_schur.hip -- MFMA Schur-complement variant of the batched node-AVI solver (gfx950).
//
// For the item shape the hot path is made of -- a node's reduced KKT system
//     kinds = [STD free x n | GAVI x m],  n, m <= 32      (M = [[H, C],[A, D]], q = [g; b])
// -- the crash (Stage A of DESIGN.md section 3: all n free variables enter) is block Gaussian
// elimination of the H block.  Here it runs as 8 rank-4 block pivots on fp64 matrix cores:
//   * the 64 x 64 (padded) matrix lives in registers as 4 x 4 tiles of 16 x 16 in the C/D layout of
//     v_mfma_f64_16x16x4_f64 (lane l, reg g  <->  row (l>>4) + 4g, col l&15); an aligned group of 4
//     rows IS a B operand, so the pivot rows need no data movement at all;
//   * per block pivot: the 4 pivot columns go through LDS once (-> A operands), the 4 x 4 pivot block is
//     inverted redundantly by every lane (uniform), V' = P^-1 V is one MFMA per column tile, and the
//     rank-4 update of a tile is ONE instruction (1024 multiply-adds) instead of 4 x 16 v_fma_f64
//     plus their LDS traffic: ~140 MFMAs replace ~8000 VALU/LDS/SALU instructions of the scalar crash.
// Result: S = D - A H^-1 C (m x m), c = b - A H^-1 g, W = H^-1 C, h = H^-1 g.  Stage B (Lemke) then
// runs on the 32 x 33 dictionary of S in the same tile layout with v_fma_f64 (a pivot touches 16
// registers per lane), and x = -(W lambda + h) is recovered at the end.  Post-check, residual and
// active-set masks are computed on the ORIGINAL blocks exactly as in qpn_avi_reg.hip.
//
// Items that do not have this shape, or whose H block fails the no-pivoting test
// (|pivot| >= 1e-4 max(1, max|M|) inside a 4 x 4 block), are flagged (status = -1) and solved by the
// register kernel in a second, gated launch -- results identical to the general path.
// Arithmetic differs from the scalar crash only by summation order (block elimination), so primals
// agree to ~1e-13 and active sets are identical on well-posed items; parity bar: DESIGN.md section 2.
#include "qpn_internal.h"

#define QINF __builtin_huge_val()

namespace {

constexpr int WAVE = 64;

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

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)

// 16 tiles, named (no arrays: see qpn_avi_reg.hip)
#define TL(I, J) tl_##I##_##J
#define FOR_J(M, I) M(I, 0) M(I, 1) M(I, 2) M(I, 3)
#define FOR_IJ(M) FOR_J(M, 0) FOR_J(M, 1) FOR_J(M, 2) FOR_J(M, 3)

struct SchurDebug { double *S, *c, *W, *h; };

// permuted position of row r (0..31) in the Stage-B column vector: lane group q = r&3 reads its 8 rows
// (r = q + 4g + 16Ib) as 8 consecutive doubles
__device__ __forceinline__ int perm32(int r) { return (r & 3) * 8 + ((r >> 4) << 2) + ((r >> 2) & 3); }

template <bool NODES>
__global__ __launch_bounds__(WAVE, 2) void avi_solve_schur(AviBatchArgs a, SchurDebug dbg)
{
    const int N = NODES ? a.nd.n + a.nd.m : a.N;
    const int l = threadIdx.x;
    const int b = blockIdx.x;
    const int lc = l & 15, lq = l >> 4;

    __shared__ __attribute__((aligned(16))) double sU[64 * 4];      // pivot columns, [row][k]
    __shared__ __attribute__((aligned(16))) double sP[16];          // P^-1, [i][k]
    __shared__ __attribute__((aligned(16))) double sucol[40];       // Stage B: pivot column (permuted) + extra
    __shared__ __attribute__((aligned(16))) double svrow[40];       // Stage B: pivot row
    __shared__ double sl[32], su[32], sval[2 * 32 + 2], sz[64];   // sval: values by variable id
    __shared__ int sat[32];

    const double *Mg = NODES ? nullptr : a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)N;
    const bool act = l < N;
#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    // node records (fused path): M = [[Qd, -Ad'],[Ad, 0]], q = [qd + R w; B w], src/avi.jl:205-251 + :305-377
    const int nn = a.nd.n, nm = a.nd.m, np_ = a.nd.p;
    const double *Q_ = NODES ? a.nd.Qd + (size_t)b * nn * nn : nullptr;
    const double *A_ = NODES ? a.nd.Ad + (size_t)b * nm * nn : nullptr;
    const double *R_ = NODES ? a.nd.R + (size_t)b * nn * np_ : nullptr;
    const double *B_ = NODES ? a.nd.B + (size_t)b * nm * np_ : nullptr;
    const double *w_ = NODES ? a.nd.w + (size_t)b * (size_t)a.nd.stride_w : nullptr;
    // element (ri, ci) of the item's stacked M, item coordinates
    auto melem = [&](int ri, int ci) -> double {
        if constexpr (NODES) {
            if (ri < nn) return ci < nn ? Q_[(size_t)ci * nn + ri] : -A_[(size_t)ri * nm + (ci - nn)];
            return ci < nn ? A_[(size_t)ci * nm + (ri - nn)] : 0.0;
        } else {
            return Mg[(size_t)ci * N + ri];
        }
    };
    auto qelem = [&](int it) -> double {
        if constexpr (NODES) {
            double s;
            if (it < nn) { s = a.nd.qd[(size_t)b * nn + it]; for (int k = 0; k < np_; ++k) s = fma(R_[(size_t)k * nn + it], w_[k], s); }
            else { s = 0.0; for (int k = 0; k < np_; ++k) s = fma(B_[(size_t)k * nm + (it - nn)], w_[k], s); }
            return s;
        } else {
            return a.q[vo + it];
        }
    };

    // ---- structure test: leading free STD rows, then GAVI rows -------------------------------------
    double lk = 0.0, uk = 0.0;
    int gk = 0;
    if constexpr (NODES) {
        if (act) {
            if (l < nn) { lk = -QINF; uk = QINF; }
            else { lk = a.nd.l[(size_t)b * nm + (l - nn)]; uk = a.nd.u[(size_t)b * nm + (l - nn)]; gk = 1; }
        }
    } else {
        lk = act ? a.l[vo + l] : 0.0; uk = act ? a.u[vo + l] : 0.0;
        gk = (act && a.kind) ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + l] : 0;
    }
    const bool isfree = act && !gk && lk == -QINF && uk == QINF;
    const unsigned long long mfree = __ballot(isfree), mg = __ballot(act && gk);
    const int n = __popcll(mfree), m = __popcll(mg);
    const bool shape_ok = n + m == N && n <= 32 && m <= 32 && n >= 1 &&
                          mfree == ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)) &&
                          mg == ((((m + n) >= 64) ? ~0ull : ((1ull << (m + n)) - 1ull)) & ~((1ull << n) - 1ull));
    if (!shape_ok) { if (l == 0) a.status[b] = -1; return; }

    // internal index (0..63) -> item index or -1 (padding): x block 0..31, lambda block 32..63
    auto item_of = [&](int r) -> int { return r < 32 ? (r < n ? r : -1) : (r - 32 < m ? n + (r - 32) : -1); };

    // ---- load straight into the MFMA tile layout: all 64 loads in flight at once ----------------------
#define M_DECL(I, J) d4 TL(I, J);
    FOR_IJ(M_DECL)
#undef M_DECL
    double mabs = 0.0;
#define M_LOAD(I, J)                                                                                \
    {                                                                                               \
        const int cc = 16 * (J) + lc;                                                               \
        const int ci = item_of(cc);                                                                 \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
            const int rr = 16 * (I) + 4 * g + lq;                                                   \
            const int ri = item_of(rr);                                                             \
            const bool valid = ri >= 0 && ci >= 0;                                                  \
            double v;                                                                               \
            if constexpr (NODES) {                                                                  \
                /* the tile's quadrant is static; loads are unconditional on a clamped index */     \
                if constexpr ((I) < 2 && (J) < 2) v = Q_[valid ? (size_t)ci * nn + ri : 0];         \
                else if constexpr ((I) < 2) v = -A_[valid ? (size_t)ri * nm + (ci - nn) : 0];       \
                else if constexpr ((J) < 2) v = A_[valid ? (size_t)ci * nm + (ri - nn) : 0];        \
                else v = 0.0;                                                                       \
            } else {                                                                                \
                v = Mg[valid ? (size_t)ci * N + ri : 0];                                            \
            }                                                                                       \
            if (!valid) v = (rr == cc && rr < 32) ? 1.0 : 0.0;      /* padded x rows: identity */     \
            if constexpr (NODES && (I) >= 2 && (J) >= 2) v = 0.0;                                   \
            TL(I, J)[g] = v;                                                                        \
            mabs = fmax(mabs, fabs(v));                                                             \
        }                                                                                           \
    }
    FOR_IJ(M_LOAD)
#undef M_LOAD
    // extra column: q in internal order, one entry per internal row (lane l <-> internal row l)
    double kx;
    {
        const int it = item_of(l);
        kx = it >= 0 ? qelem(it) : 0.0;
    }
    const double mscale = wave_max_f64(mabs);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);

    STAMP(0);   // setup + load
    // ---- Stage A: 8 rank-4 block pivots on the matrix cores --------------------------------------------
    bool fail = false;
#define M_GATHER(I, JP)                                                                             \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq;                                                       \
        double v = TL(I, JP)[g];                                                                    \
        if (rr == p0 + kcol) v -= 1.0;              /* pivot rows carry P - I (see header) */        \
        sU[rr * 4 + kcol] = v;                                                                      \
    }
#define M_UPD(I, J) TL(I, J) = MFMA(au##I, vn, TL(I, J));
#define M_COLTILE(J, IP, GP)                                                                        \
    {                                                                                               \
        const double vraw = TL(IP, J)[GP];                                                          \
        d4 z4 = {0.0, 0.0, 0.0, 0.0};                                                               \
        const d4 vn4 = MFMA(ap, vraw, z4);                                                          \
        const double vn = vn4[0];                                                                   \
        M_UPD(0, J) M_UPD(1, J) M_UPD(2, J) M_UPD(3, J)                                             \
    }
#define M_STEP(KB, JP, GP)                                                                          \
    if (!fail) {                                                                                    \
        constexpr int p0 = 4 * (KB);                                                                \
        constexpr int cq = p0 & 15;                                                                 \
        const int kcol = lc - cq;                                                                   \
        if (kcol >= 0 && kcol < 4) { M_GATHER(0, JP) M_GATHER(1, JP) M_GATHER(2, JP) M_GATHER(3, JP) } \
        __syncthreads();                                                                            \
        /* P = pivot block (+I back), inverted by every lane (uniform values) */                    \
        double p00 = sU[(p0 + 0) * 4 + 0] + 1.0, p01 = sU[(p0 + 0) * 4 + 1], p02 = sU[(p0 + 0) * 4 + 2], p03 = sU[(p0 + 0) * 4 + 3]; \
        double p10 = sU[(p0 + 1) * 4 + 0], p11 = sU[(p0 + 1) * 4 + 1] + 1.0, p12 = sU[(p0 + 1) * 4 + 2], p13 = sU[(p0 + 1) * 4 + 3]; \
        double p20 = sU[(p0 + 2) * 4 + 0], p21 = sU[(p0 + 2) * 4 + 1], p22 = sU[(p0 + 2) * 4 + 2] + 1.0, p23 = sU[(p0 + 2) * 4 + 3]; \
        double p30 = sU[(p0 + 3) * 4 + 0], p31 = sU[(p0 + 3) * 4 + 1], p32 = sU[(p0 + 3) * 4 + 2], p33 = sU[(p0 + 3) * 4 + 3] + 1.0; \
        double q00 = 1, q01 = 0, q02 = 0, q03 = 0, q10 = 0, q11 = 1, q12 = 0, q13 = 0;              \
        double q20 = 0, q21 = 0, q22 = 1, q23 = 0, q30 = 0, q31 = 0, q32 = 0, q33 = 1;              \
        bool okp = fabs(p00) >= diag_thr;                                                           \
        { const double iv = 1.0 / p00; p01 *= iv; p02 *= iv; p03 *= iv; q00 *= iv;                  \
          { const double f = p10; p11 -= f * p01; p12 -= f * p02; p13 -= f * p03; q10 -= f * q00; } \
          { const double f = p20; p21 -= f * p01; p22 -= f * p02; p23 -= f * p03; q20 -= f * q00; } \
          { const double f = p30; p31 -= f * p01; p32 -= f * p02; p33 -= f * p03; q30 -= f * q00; } } \
        okp = okp && fabs(p11) >= diag_thr;                                                         \
        { const double iv = 1.0 / p11; p12 *= iv; p13 *= iv; q10 *= iv; q11 *= iv;                  \
          { const double f = p01; p02 -= f * p12; p03 -= f * p13; q00 -= f * q10; q01 -= f * q11; } \
          { const double f = p21; p22 -= f * p12; p23 -= f * p13; q20 -= f * q10; q21 -= f * q11; } \
          { const double f = p31; p32 -= f * p12; p33 -= f * p13; q30 -= f * q10; q31 -= f * q11; } } \
        okp = okp && fabs(p22) >= diag_thr;                                                         \
        { const double iv = 1.0 / p22; p23 *= iv; q20 *= iv; q21 *= iv; q22 *= iv;                  \
          { const double f = p02; p03 -= f * p23; q00 -= f * q20; q01 -= f * q21; q02 -= f * q22; } \
          { const double f = p12; p13 -= f * p23; q10 -= f * q20; q11 -= f * q21; q12 -= f * q22; } \
          { const double f = p32; p33 -= f * p23; q30 -= f * q20; q31 -= f * q21; q32 -= f * q22; } } \
        okp = okp && fabs(p33) >= diag_thr;                                                         \
        { const double iv = 1.0 / p33; q30 *= iv; q31 *= iv; q32 *= iv; q33 *= iv;                  \
          { const double f = p03; q00 -= f * q30; q01 -= f * q31; q02 -= f * q32; q03 -= f * q33; } \
          { const double f = p13; q10 -= f * q30; q11 -= f * q31; q12 -= f * q32; q13 -= f * q33; } \
          { const double f = p23; q20 -= f * q30; q21 -= f * q31; q22 -= f * q32; q23 -= f * q33; } } \
        if (!ubool(okp)) { fail = true; }                                                           \
        else {                                                                                      \
            if (l == 0) {                                                                           \
                sP[0] = q00; sP[1] = q01; sP[2] = q02; sP[3] = q03; sP[4] = q10; sP[5] = q11; sP[6] = q12; sP[7] = q13; \
                sP[8] = q20; sP[9] = q21; sP[10] = q22; sP[11] = q23; sP[12] = q30; sP[13] = q31; sP[14] = q32; sP[15] = q33; \
            }                                                                                       \
            /* extra column: kx_i -= sum_k U[i][k] (P^-1 kx_piv)[k]   (lane l <-> internal row l) */  \
            {                                                                                       \
                const double x0 = readlane_f64(kx, p0 + 0), x1 = readlane_f64(kx, p0 + 1);          \
                const double x2 = readlane_f64(kx, p0 + 2), x3 = readlane_f64(kx, p0 + 3);          \
                const double y0 = q00 * x0 + q01 * x1 + q02 * x2 + q03 * x3;                        \
                const double y1 = q10 * x0 + q11 * x1 + q12 * x2 + q13 * x3;                        \
                const double y2 = q20 * x0 + q21 * x1 + q22 * x2 + q23 * x3;                        \
                const double y3 = q30 * x0 + q31 * x1 + q32 * x2 + q33 * x3;                        \
                kx -= sU[l * 4 + 0] * y0 + sU[l * 4 + 1] * y1 + sU[l * 4 + 2] * y2 + sU[l * 4 + 3] * y3; \
            }                                                                                       \
            __syncthreads();                                                                        \
            const double ap = lc < 4 ? sP[lc * 4 + lq] : 0.0;       /* A operand: P^-1 padded to 16 x 4 */ \
            const double au0 = -sU[(0 + lc) * 4 + lq], au1 = -sU[(16 + lc) * 4 + lq];               \
            const double au2 = -sU[(32 + lc) * 4 + lq], au3 = -sU[(48 + lc) * 4 + lq];              \
            if ((JP) <= 0) M_COLTILE(0, JP, GP)                                                     \
            if ((JP) <= 1) M_COLTILE(1, JP, GP)                                                     \
            M_COLTILE(2, JP, GP)                                                                    \
            M_COLTILE(3, JP, GP)                                                                    \
            __syncthreads();                                                                        \
        }                                                                                           \
    }
    M_STEP(0, 0, 0) M_STEP(1, 0, 1) M_STEP(2, 0, 2) M_STEP(3, 0, 3)
    M_STEP(4, 1, 0) M_STEP(5, 1, 1) M_STEP(6, 1, 2) M_STEP(7, 1, 3)
#undef M_STEP
#undef M_COLTILE
#undef M_UPD
#undef M_GATHER
    if (fail) { if (l == 0) a.status[b] = -1; return; }
    STAMP(6);   // crash on the matrix cores

    if (dbg.S) {
        // diagnostic builds: dump S (32x32), c, W (32x32), h in row-major
#define M_DUMP(I, J)                                                                                \
    if ((J) >= 2) {                                                                                 \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
            const int rr = 16 * (I) + 4 * g + lq, cc = 16 * ((J) - 2) + lc;                         \
            double *dst = (I) >= 2 ? dbg.S : dbg.W;                                                 \
            dst[(size_t)b * 1024 + (size_t)(rr & 31) * 32 + cc] = TL(I, J)[g];                      \
        }                                                                                           \
    }
        FOR_IJ(M_DUMP)
#undef M_DUMP
        if (l < 32) dbg.h[(size_t)b * 32 + l] = kx; else dbg.c[(size_t)b * 32 + (l - 32)] = kx;
        if (l == 0) a.status[b] = -2;
        return;
    }

    // ================= Stage B: Lemke on the Schur dictionary (32 pairs, tile layout) =================
    // pair k (k < 32) <-> item row n + k:  p_k = s_k = (S lambda + c)_k in [l_k, u_k],  d_k = lambda_k.
    // ids: p_k -> k, d_k -> 32 + k, artificial -> 64; column index 32 = the extra (covering) column.
    constexpr int NBP = 32, XC = 32, VTH = 64;
    const bool actb = l < NBP;
    double xb = __shfl(kx, (l + 32) & 63, WAVE);            // c_k sits in lane 32 + k
    double lo = -QINF, hi = QINF;
    {
        const int it = l < m ? n + l : -1;
        if (actb && it >= 0) {
            if constexpr (NODES) { lo = a.nd.l[(size_t)b * nm + l]; hi = a.nd.u[(size_t)b * nm + l]; }
            else { lo = a.l[vo + it]; hi = a.u[vo + it]; }
        }
        if (actb) { sl[l] = lo; su[l] = hi; sat[l] = 0; }
    }
    // equality GAVI rows need their multiplier crashed in: left to the general kernel
    if (__ballot(actb && lo == hi)) { if (l == 0) a.status[b] = -1; return; }
    int rowvar = actb ? l : -1, colvar = actb ? NBP + l : -1;
    double nbval = 0.0, tcol = 0.0;
    int cNvar = VTH;
    double cNval = 0.0;
    __syncthreads();

#define SB(Ib, Jb) TL_S_##Ib##_##Jb
#define TL_S_0_0 TL(2, 2)
#define TL_S_0_1 TL(2, 3)
#define TL_S_1_0 TL(3, 2)
#define TL_S_1_1 TL(3, 3)
    auto col_of = [&](int v) -> int {
        int cc = wave_first(actb && colvar == v);
        if (cc < 0 && cNvar == v) cc = XC;
        return cc;
    };

    int pivots = n;                       // the crash brought n free variables in (Stage A)
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;
    int status = QPN_FAILURE;
    int c = XC;
    double sigma = -1.0, self_lim = 0.0, elo = 0.0, ehi = QINF;
    const double slack = 1e-10, ptol = a.piv_tol;
    {
        double viol = 0.0;
        if (actb) viol = xb < lo ? lo - xb : (xb > hi ? xb - hi : 0.0);
        const double theta0 = wave_max_f64(viol);
        if (ubool(theta0 <= a.feas_tol)) status = QPN_SUCCESS;
        else {
            if (actb) {
                double cov = 0.0;
                if (xb < lo) {
                    double tgt = lo + (theta0 - (lo - xb));
                    if (hi < QINF) { double mid = 0.5 * (lo + hi); if (tgt > mid) tgt = mid; }
                    cov = (tgt - xb) / theta0; xb = tgt;
                } else if (xb > hi) {
                    double tgt = hi - (theta0 - (xb - hi));
                    if (lo > -QINF) { double mid = 0.5 * (lo + hi); if (tgt < mid) tgt = mid; }
                    cov = (tgt - xb) / theta0; xb = tgt;
                }
                tcol = cov;
            }
            cNval = theta0; self_lim = theta0;
            status = QPN_MAX_ITERS;
        }
    }
    while (status == QPN_MAX_ITERS) {
        if (pivots >= max_piv) break;
        c = uni(c);
        // ---- entering column -> sucol (permuted so a lane group reads its 8 rows contiguously)
        if (c == XC) { if (actb) sucol[perm32(l)] = tcol; }
        else if (lc == (c & 15)) {
            if ((c >> 4) == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) { sucol[lq * 8 + g] = SB(0, 0)[g]; sucol[lq * 8 + 4 + g] = SB(1, 0)[g]; }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) { sucol[lq * 8 + g] = SB(0, 1)[g]; sucol[lq * 8 + 4 + g] = SB(1, 1)[g]; }
            }
        }
        __syncthreads();
        const double cm = actb ? sucol[perm32(l)] : 0.0;
        // ---- ratio test (same rule and arithmetic as the general kernel)
        const double gdir = sigma * cm;
        const double rc = 1.0 / gdir;
        const bool cndlo = actb && gdir < -ptol && lo > -QINF;
        const bool cndhi = actb && gdir > ptol && hi < QINF;
        const bool cnd = cndlo || cndhi;
        const double arc = cndlo ? -rc : rc;
        const double dd = (cndlo ? xb - lo : hi - xb) * arc;
        const double d1 = dd + slack * arc;
        double dmax = wave_min_f64(cnd ? d1 : QINF);
        if (self_lim < dmax) dmax = self_lim;
        dmax = udbl(dmax);
        if (ubool(dmax == QINF)) { status = QPN_RAY_TERM; break; }
        const bool cand = cnd && dd <= dmax;
        const unsigned long long bal = __ballot(cand);
        if (bal == 0ull) {
            const double dl = sigma * self_lim;
            if (actb) xb = fma(dl, cm, xb);
            const int ve = (c == XC) ? cNvar : readlane_i32(colvar, c);
            if (ve == VTH) {
                if (c == XC) cNval = 0.0; else if (l == c) nbval = 0.0;
                status = QPN_SUCCESS; break;
            }
            const int k = ve;
            const int au = ubool(sigma > 0.0) ? 1 : 0;
            const double nv = udbl(au ? su[k] : sl[k]);
            if (l == 0) sat[k] = au;
            if (c == XC) cNval = nv; else if (l == c) nbval = nv;
            pivots++;
            c = col_of(NBP + k);
            if (c < 0) { status = QPN_FAILURE; break; }
            sigma = au ? -1.0 : 1.0;
            self_lim = QINF;
            if (au) { elo = -QINF; ehi = 0.0; } else { elo = 0.0; ehi = QINF; }
            __syncthreads();
            continue;
        }
        int r;
        if (__popcll(bal) == 1) r = __ffsll((long long)bal) - 1;
        else {
            double ag = cand ? fabs(gdir) : -1.0;
            if (cand && rowvar == VTH) ag = QINF;
            const double bestg = wave_max_f64(ag);
            r = wave_first(cand && ag == bestg);
        }
        r = uni(r);
        double step = readlane_f64(dd, r);
        if (ubool(step < 0.0)) step = 0.0;
        const double leave_val = udbl(readlane_f64(cndlo ? lo : hi, r));
        const double inv = udbl(sigma * readlane_f64(rc, r));
        const double delta = udbl(sigma * step);
        const int vl = readlane_i32(rowvar, r);
        // ---- pivot: row r -> svrow (raw), then the rank-1 update on the four S tiles
        const double enter_val = udbl(((c == XC) ? cNval : readlane_f64(nbval, c)) + delta);
        const int rq = r & 3, rsel = ((r >> 4) << 2) | ((r >> 2) & 3);      // leaf = Ib*4 + g
        if (lq == rq) {
#define M_XROW(Ib, g) { svrow[lc] = SB(Ib, 0)[g]; svrow[16 + lc] = SB(Ib, 1)[g]; }
            if (rsel < 4) { if (rsel < 2) { if (rsel < 1) M_XROW(0, 0) else M_XROW(0, 1) } else { if (rsel < 3) M_XROW(0, 2) else M_XROW(0, 3) } }
            else { if (rsel < 6) { if (rsel < 5) M_XROW(1, 0) else M_XROW(1, 1) } else { if (rsel < 7) M_XROW(1, 2) else M_XROW(1, 3) } }
#undef M_XROW
        }
        if (l == r) svrow[32] = tcol;
        __syncthreads();
        {
            const double vxr = svrow[32];
            double u0 = sucol[lq * 8 + 0], u1 = sucol[lq * 8 + 1], u2 = sucol[lq * 8 + 2], u3 = sucol[lq * 8 + 3];
            double u4 = sucol[lq * 8 + 4], u5 = sucol[lq * 8 + 5], u6 = sucol[lq * 8 + 6], u7 = sucol[lq * 8 + 7];
            double v0 = svrow[lc] * inv, v1 = svrow[16 + lc] * inv;
            if (lc == c) v0 = -inv;
            if (16 + lc == c) v1 = -inv;
            {
                const double vx = (c == XC) ? -inv : vxr * inv;
                double xbn = fma(delta, cm, xb);
                double tcn = (c == XC) ? cm * inv : fma(-cm, vx, tcol);
                if (l == r) { xbn = enter_val; tcn = (c == XC) ? inv : -vx; }
                xb = xbn; tcol = tcn;
            }
            SB(0, 0)[0] = fma(-u0, v0, SB(0, 0)[0]); SB(0, 0)[1] = fma(-u1, v0, SB(0, 0)[1]);
            SB(0, 0)[2] = fma(-u2, v0, SB(0, 0)[2]); SB(0, 0)[3] = fma(-u3, v0, SB(0, 0)[3]);
            SB(0, 1)[0] = fma(-u0, v1, SB(0, 1)[0]); SB(0, 1)[1] = fma(-u1, v1, SB(0, 1)[1]);
            SB(0, 1)[2] = fma(-u2, v1, SB(0, 1)[2]); SB(0, 1)[3] = fma(-u3, v1, SB(0, 1)[3]);
            SB(1, 0)[0] = fma(-u4, v0, SB(1, 0)[0]); SB(1, 0)[1] = fma(-u5, v0, SB(1, 0)[1]);
            SB(1, 0)[2] = fma(-u6, v0, SB(1, 0)[2]); SB(1, 0)[3] = fma(-u7, v0, SB(1, 0)[3]);
            SB(1, 1)[0] = fma(-u4, v1, SB(1, 1)[0]); SB(1, 1)[1] = fma(-u5, v1, SB(1, 1)[1]);
            SB(1, 1)[2] = fma(-u6, v1, SB(1, 1)[2]); SB(1, 1)[3] = fma(-u7, v1, SB(1, 1)[3]);
            // column c: T[i][c] = cm_i * inv   (the 4 lanes that own that column)
            if (c != XC && lc == (c & 15)) {
                if ((c >> 4) == 0) {
                    SB(0, 0)[0] = u0 * inv; SB(0, 0)[1] = u1 * inv; SB(0, 0)[2] = u2 * inv; SB(0, 0)[3] = u3 * inv;
                    SB(1, 0)[0] = u4 * inv; SB(1, 0)[1] = u5 * inv; SB(1, 0)[2] = u6 * inv; SB(1, 0)[3] = u7 * inv;
                } else {
                    SB(0, 1)[0] = u0 * inv; SB(0, 1)[1] = u1 * inv; SB(0, 1)[2] = u2 * inv; SB(0, 1)[3] = u3 * inv;
                    SB(1, 1)[0] = u4 * inv; SB(1, 1)[1] = u5 * inv; SB(1, 1)[2] = u6 * inv; SB(1, 1)[3] = u7 * inv;
                }
            }
            // row r: T[r][j] = -prow_j, T[r][c] = inv (v carries -inv there)
            if (lq == rq) {
#define M_FROW(Ib, g) { SB(Ib, 0)[g] = -v0; SB(Ib, 1)[g] = -v1; }
                if (rsel < 4) { if (rsel < 2) { if (rsel < 1) M_FROW(0, 0) else M_FROW(0, 1) } else { if (rsel < 3) M_FROW(0, 2) else M_FROW(0, 3) } }
                else { if (rsel < 6) { if (rsel < 5) M_FROW(1, 0) else M_FROW(1, 1) } else { if (rsel < 7) M_FROW(1, 2) else M_FROW(1, 3) } }
#undef M_FROW
            }
        }
        {
            const int ve = (c == XC) ? cNvar : readlane_i32(colvar, c);
            if (l == r) { rowvar = ve; lo = elo; hi = ehi; }
            if (c == XC) { cNvar = vl; cNval = leave_val; }
            else if (l == c) { colvar = vl; nbval = leave_val; }
        }
        pivots++;
        if (vl == VTH) { status = QPN_SUCCESS; break; }
        int vn;
        if (vl < NBP) {
            const int k = vl;
            const double Lk = udbl(sl[k]), Uk = udbl(su[k]);
            int au = uni(sat[k]);
            if (ubool(Lk != Uk)) { au = (leave_val == Uk) ? 1 : 0; if (l == 0) sat[k] = au; }
            vn = NBP + k;
            sigma = au ? -1.0 : 1.0;
            self_lim = QINF;
            if (ubool(Lk == Uk)) { elo = -QINF; ehi = QINF; }
            else if (ubool(Lk == -QINF && Uk == QINF)) { elo = 0.0; ehi = 0.0; }
            else if (au) { elo = -QINF; ehi = 0.0; }
            else { elo = 0.0; ehi = QINF; }
        } else {
            const int k = vl - NBP;
            const double Lk = udbl(sl[k]), Uk = udbl(su[k]);
            vn = k;
            sigma = uni(sat[k]) ? -1.0 : 1.0;
            self_lim = Uk - Lk;
            if (ubool(Lk == -QINF && Uk == QINF)) { self_lim = QINF; sigma = 1.0; }
            elo = Lk; ehi = Uk;
        }
        c = col_of(vn);
        if (c < 0) { status = QPN_FAILURE; break; }
        __syncthreads();
    }

    STAMP(4);   // Lemke (all phases)
    // ---- read back: lambda_k, then x = -(W lambda + h) ---------------------------------------------------
    __syncthreads();
    if (actb) { sval[rowvar] = xb; sval[colvar] = nbval; }
    if (l == 0) sval[cNvar] = cNval;
    __syncthreads();
    const double lam0 = sval[NBP + lc], lam1 = sval[NBP + 16 + lc];     // lambda of this lane's two columns
    {
        // partial row sums of W lambda over this lane's columns, reduced across the 16 lanes of a DPP row
#define M_WROW(Ib, g, slot)                                                                         \
    {                                                                                               \
        double pr = TL(Ib, 2)[g] * lam0 + TL(Ib, 3)[g] * lam1;                                      \
        pr += dpp_f64<0xB1>(pr); pr += dpp_f64<0x4E>(pr); pr += dpp_f64<0x141>(pr); pr += dpp_f64<0x140>(pr); \
        if (lc == 0) sz[16 * (Ib) + 4 * (g) + lq] = pr;                                             \
    }
        M_WROW(0, 0, 0) M_WROW(0, 1, 1) M_WROW(0, 2, 2) M_WROW(0, 3, 3)
        M_WROW(1, 0, 4) M_WROW(1, 1, 5) M_WROW(1, 2, 6) M_WROW(1, 3, 7)
#undef M_WROW
    }
    __syncthreads();
    // item order: rows < n are x, rows n.. are lambda
    double zk = 0.0;
    if (act) zk = l < n ? -(sz[l] + kx) : sval[NBP + (l - n)];
    __syncthreads();
    if (act) sz[l] = zk;
    __syncthreads();

    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -------------------------------
    double rk = act ? qelem(l) : 0.0;
    if constexpr (NODES) {
        // r = q + M z from the node records, item columns in ascending order; 8 loads in flight per
        // step, a zero z_j contributes exactly nothing (finite blocks)
        const bool isx = l < nn;
        const int ls = act ? l : 0;
        const double *colb = isx ? Q_ + ls : A_ + (act ? ls - nn : 0);      // column sweep base of this row
        const size_t cst = isx ? (size_t)nn : (size_t)nm;
        const double *rowb = A_ + (size_t)(isx ? ls : 0) * nm;                // row of A' for x rows
        int j = 0;
        for (; j + 8 <= nn; j += 8) {                     // columns of x: Q (x rows) or A (constraint rows)
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mv[q8] = colb[(size_t)(j + q8) * cst];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { const double zj = sz[j + q8]; rk = (zj != 0.0 && act) ? fma(mv[q8], zj, rk) : rk; }
        }
        for (; j < nn; ++j) {
            const double zj = sz[j];
            if (zj != 0.0 && act) rk = fma(colb[(size_t)j * cst], zj, rk);
        }
        int k = 0;
        for (; k + 8 <= nm; k += 8) {                     // columns of lambda: -A' (x rows only)
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mv[q8] = -rowb[k + q8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { const double zj = sz[nn + k + q8]; rk = (zj != 0.0 && act && isx) ? fma(mv[q8], zj, rk) : rk; }
        }
        for (; k < nm; ++k) {
            const double zj = sz[nn + k];
            if (zj != 0.0 && act && isx) rk = fma(-rowb[k], zj, rk);
        }
    } else {
        int j = 0;
        for (; j + 8 <= N; j += 8) {
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mv[q8] = act ? Mg[(size_t)(j + q8) * N + l] : 0.0;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { const double zj = sz[j + q8]; rk = (zj != 0.0) ? fma(mv[q8], zj, rk) : rk; }
        }
        for (; j < N; ++j) { const double zj = sz[j]; if (zj != 0.0 && act) rk = fma(Mg[(size_t)j * N + l], zj, rk); }
    }
    const double pp = gk ? rk : zk, dv = gk ? zk : rk;
    int bad = 0;
    double nres = 0.0;
    unsigned mask = 0;
    if (act) {
        const double tol = a.check_tol;
        if (dv > tol && fabs(pp - lk) > tol) bad++;
        if (dv < -tol && fabs(pp - uk) > tol) bad++;
        if (pp - lk < -tol) bad++;
        if (pp - uk > tol) bad++;
        if (isnan(pp) || isnan(dv)) bad++;
        double tt = pp - dv;
        if (tt < lk) tt = lk;
        if (tt > uk) tt = uk;
        nres = fabs(pp - tt);
        if (isnan(nres)) nres = QINF;
        const double ct = a.comp_tol;
        auto approx = [&](double x, double y) { return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= ct); };
        if (!approx(lk, uk)) {
            if (approx(pp, lk) && dv >= -ct) mask |= 1u;
            if (lk - ct <= pp && pp <= uk + ct && fabs(dv) <= ct) mask |= 2u;
            if (approx(pp, uk) && dv <= ct) mask |= 4u;
        } else mask = 8u;
        if (gk) mask <<= 4;
    }
    bad = wave_sum_i32(bad);
    nres = wave_max_f64(nres);
    if (bad > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
    if (act) {
        a.z[vo + l] = zk;
        if (a.active) a.active[vo + l] = (uint8_t)mask;
    }
    if (l == 0) {
        a.status[b] = status;
        if (a.resid) a.resid[b] = nres;
        if (a.pivots) a.pivots[b] = pivots;
    }
    STAMP(5);   // read-back + post-check + stores
#ifdef QPN_STAMPS
    if (a.stamps && l == 0)
        for (int i = 0; i < 8; ++i) a.stamps[(size_t)b * 8 + i] = stamp_acc[i];
#endif
}

} // namespace

hipError_t qpn_launch_avi_solve_schur(const AviBatchArgs &a, double *dbgS, double *dbgc, double *dbgW,
                                      double *dbgh, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    SchurDebug d{dbgS, dbgc, dbgW, dbgh};
    hipLaunchKernelGGL(avi_solve_schur<false>, dim3((unsigned)a.batch), dim3(WAVE), 0, stream, a, d);
    return hipGetLastError();
}

hipError_t qpn_launch_avi_solve_schur_nodes(const AviBatchArgs &a, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    SchurDebug d{nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(avi_solve_schur<true>, dim3((unsigned)a.batch), dim3(WAVE), 0, stream, a, d);
    return hipGetLastError();
}
