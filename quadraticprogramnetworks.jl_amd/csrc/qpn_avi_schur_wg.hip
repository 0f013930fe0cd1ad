// qpn_avi_schur_wg.hip -- ONE fused workgroup kernel for MID-SIZE node records (n, m <= 64, one of them > 32), gfx950.
//
// Round 2 solved these nodes with three kernels (qpn_avi_schur_mid.hip: crash on the matrix cores, the Schur problem on the
// general register kernel, a finish kernel) that handed W~, S and c through an HBM workspace: ~600 MB per 4 000 nodes where
// the records and the outputs are ~150 MB, and a Lemke phase at 5.3 K clocks per pivot.  Here the whole solve of a node --
// KKT assembly (src/avi.jl:205-251, :305-377), crash, Lemke, read-back, post-check (src/avi.jl:71-76, :148-156), active
// sets (src/avi_solutions.jl:511-562), primal write-back (src/avi.jl:440-443) -- is ONE workgroup of NW = n_pad/16 (3 or 4)
// wavefronts; nothing but the records is read and nothing but the outputs is written.
//
//   Stage A (crash of the n free variables = block Gauss-Jordan on [H | C~ | g], C~ = +Ad'): wave v owns ROW TILE v
//     (16 rows x up to 8 MFMA tiles = 64 VGPRs).  Per rank-4 block pivot (n/4 <= 16 of them) the owner of the pivot rows
//     publishes them raw (B operands) and the raw 4 x 4 pivot block through LDS, every wave gathers its 16 x 4 panel of pivot
//     columns: ONE workgroup barrier; then EVERY wave factors the pivot block itself (uniform, in registers: 4 x 4 LU without
//     pivoting, every pivot must pass |u_ss| >= 1e-4 max(1, max|M|), else the node is declined) and forms its entry of
//     U' = U P^-1 in the A-operand layout: no second hand-over, one MFMA (NEG on A) per live tile.
//   S = Ad W~ and c = b - Ad h: W~ crosses LDS once (it STAYS in the tile registers for the read-back), wave v computes row
//     tile v of S; the tiles change hands through LDS (row tiles -> column tiles) for Stage B.
//   Stage B (Lemke on the m x (m+1) Schur dictionary): wave v holds COLUMN TILE v (all rows, columns 16v .. 16v+15) in the
//     MFMA tile layout (4 MT entries per lane), as the 32-class kernel holds its 32 x 32 dictionary.  Wave 0 is the leader:
//     it alone keeps the bookkeeping and runs the ratio test on the entering column, which the 4 lanes that hold it publish
//     through LDS; the other waves wait for its decision {row, 1 / pivot, next column}: two barriers per pivot.  The rank-1
//     exchange of a wave is 4 MT v_fma_f64 plus lane-masked fix-ups of the pivot column and row -- its piece of the pivot row
//     travels through a wave-private LDS vector, no cross-lane instruction -- so a pivot costs the workgroup ~100 + 35 NW
//     vector instructions instead of ~200 NW for a layout in which every wave runs the ratio test itself.
//   Read-back x = W~ lambda - h from the tile registers (DPP butterflies), post-check / residual / masks on the ORIGINAL
//     blocks: Qd re-read column-wise (coalesced, L2 / Infinity Cache), Ad staged in LDS.
// Declined nodes (a block pivot below the threshold, an equality row) keep status -1 and take the general path in gated
// launches, exactly as for the 32-class kernel.  Arithmetic differs from the scalar crash by summation order (block
// elimination) and one-step Newton reciprocals: parity bar DESIGN.md section 2.
#include "qpn_internal.h"

#define QINF __builtin_huge_val()

namespace {

constexpr int VLD = 144;                // row stride of the published pivot rows (128 columns; == 16 mod 32)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d16 __attribute__((ext_vector_type(16)));
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)
#define MFMA_NEGA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 1)      // D = C - A B (gfx950 NEG bits)


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
__device__ __forceinline__ double max_abs_nc(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min over all 64 lanes of v and the wave-uniform `lim`, returned wave-uniform: four DPP row stages, row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3, lane 63 read out (callers feed no NaNs; non-candidates carry +inf)
__device__ __forceinline__ double wave_min64_with_limit_f64(double v, double lim)
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
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xA, 0xF, false);      // rows 1, 3 <- lane 15 of rows 0, 2
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xA, 0xF, false);
        v = min_f64_nc(v, __hiloint2double(hi, lo));
    }
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x143, 0xC, 0xF, false);      // rows 2, 3 <- lane 31
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x143, 0xC, 0xF, false);
        v = min_f64_nc(v, __hiloint2double(hi, lo));
    }
    return readlane_f64(v, 63);
}

// lane id recomputed on the spot (v_mbcnt on the full EXEC mask; opaque, so the compiler does not keep the copy from the
// start of the kernel alive -- and spill it -- across the phases): every phase derives its own lane coordinates
__device__ __forceinline__ int lane_id_fresh()
{
    int x = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(x));
    return x;
}
// v + (v of lane ^ 16) + (v of lane ^ 32) + (v of lane ^ 48): gfx950's row / half swaps (VALU, no LDS trip, no address registers)
__device__ __forceinline__ double xsum_rows(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(s); hi = __double2hiint(s);
    const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
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

// LDS map (doubles).  The big area is, in turn: the Qd staging block [n_pad][n_pad + 2], the published pivot rows
// [2][4][VLD], W~ for the S product [n_pad][m_pad], S on its way into the Stage-B layout [m_pad][m_pad + 1], Ad for the
// post-check [n][m | 1].
constexpr int OFF_Q = 0;                // q = [g ; b] in item order                              [128]
constexpr int OFF_U = 128;              // Stage A: per wave its 16 x 4 panel of pivot columns    [4][64]
constexpr int OFF_PR = 384;             // Stage A: raw pivot block + x_piv, two buffers          [2][24]
constexpr int OFF_VAL = 128;            // read-back: values by variable id (over OFF_U)          [136]
constexpr int OFF_Z = 264;              // read-back: z in item order (over OFF_U / OFF_PR)       [128]
constexpr int OFF_RED = 432;            // block reduction                                        [8]
constexpr int OFF_H = 440;              // h (the eliminated extra column)                        [64]
constexpr int OFF_COL = 504;            // Stage B: entering column, two buffers; [64..127] first carries c   [2][64]
constexpr int OFF_BIG = 632;
__host__ __device__ constexpr int wg_lds_doubles(int n, int m, int n_pad, int m_pad)
{
    int big = 2 * 4 * VLD;
    if (n_pad * m_pad > big) big = n_pad * m_pad;
    if (n_pad * (n_pad + 2) > big) big = n_pad * (n_pad + 2);
    if (m_pad * (m_pad + 1) > big) big = m_pad * (m_pad + 1);
    if (n * (m | 1) > big) big = n * (m | 1);
    return OFF_BIG + big;
}

// NT, MT: the tile counts of a launch (n <= 16 NT, m <= 16 MT; sizes inside a class are padded: identity rows in H, zero
// rows / columns elsewhere); the workgroup has NW = max(NT, MT) wavefronts.  Everything that depends on the tile counts is
// resolved at compile time: the tile code is straight-line.
template <int NT, int MT>
__global__ __launch_bounds__(64 * (NT > MT ? NT : MT), 4) void schur_wg_nodes(AviBatchArgs a)
{
    constexpr int NW = NT > MT ? NT : MT;

    // schedule hint of a resident handle (longest first): this workgroup's node; a bad entry leaves the slot idle (uniform exit)
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if ((unsigned)b >= (unsigned)a.batch) return;
    const int v = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int l = (int)threadIdx.x & 63, lc = l & 15, lq = l >> 4, tid = 64 * v + l;
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    constexpr int n_pad = 16 * NT, m_pad = 16 * MT, nht = NT, mct = MT;
    extern __shared__ __attribute__((aligned(32))) double sm[];
    double *const sQ = sm + OFF_Q, *const sUv = sm + OFF_U + 64 * v, *const sPr = sm + OFF_PR;
    double *const sRed = sm + OFF_RED, *const sH = sm + OFF_H, *const sV = sm + OFF_BIG, *const sW = sm + OFF_BIG;
    double *const sval = sm + OFF_VAL, *const sz = sm + OFF_Z;

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
#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif

    // ---- load: wave v takes row tile v of [H | C~] straight into the MFMA C/D layout --------------------------
    // H(r, c) = Qd[c * n + r] (padded rows: identity), C~(r, k) = Ad[r * m + k] (Ad is m x n column-major)
    d4 th0, th1, th2, th3, tc0, tc1, tc2, tc3;
    double mabs = 0.0;
#define M_LOADC(J, T)                                                                               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * v + 4 * g + lq, ck = 16 * (J) + lc;                                     \
        const bool valid = (J) < MT && v < NT && rr < n && ck < m;                                  \
        double t_ = 0.0;                                                                            \
        if ((J) < MT) t_ = A_[valid ? (size_t)rr * m + ck : 0];                                     \
        T[g] = valid ? t_ : 0.0;                                                                    \
    }
    M_LOADC(0, tc0) M_LOADC(1, tc1) M_LOADC(2, tc2) M_LOADC(3, tc3)
#undef M_LOADC
    // Qd: whole columns with coalesced loads (lane <-> row, the NW waves take every NW-th column, all loads of a thread
    // in flight together) into LDS, column stride n_pad + 2 (conflict-free tile reads); tiles from there
    constexpr int LDQ = n_pad + 2;
    double *const sQd = sm + OFF_BIG;
    {
        double vq[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = v + NW * t;
            const bool ok = j < n && l < n;
            const double t_ = Q_[ok ? (size_t)j * n + l : 0];
            vq[t] = ok ? t_ : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = v + NW * t;
            mabs = max_abs_nc(mabs, vq[t]);
            if (j == l && l >= n) vq[t] = 1.0;                      // padded rows pivot on themselves
            if (j < n_pad && l < n_pad) sQd[j * LDQ + l] = vq[t];
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        mabs = max_abs_nc(max_abs_nc(mabs, tc0[g]), tc1[g]);
        mabs = max_abs_nc(max_abs_nc(mabs, tc2[g]), tc3[g]);
    }
    // q = [qd + R w; B w], the p terms in ascending order (the fma chain of the assembly kernel), eight loads in flight
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
    }
    // bounds of pair l (every wave keeps the whole bookkeeping of Stage B); equality GAVI rows need their multiplier
    // crashed in: left to the general kernel
    const bool actb = l < m;
    bool eqrow = false;
    if (actb) eqrow = a.nd.l[(size_t)b * m + l] == a.nd.u[(size_t)b * m + l];
    {
        double r = mabs;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(r, off, 64); r = o > r ? o : r; }
        if (l == 0) sRed[v] = r;
    }
    if (__syncthreads_or(eqrow ? 1 : 0)) { decline(); return; }
    const d4 z4 = {0.0, 0.0, 0.0, 0.0};
#define M_LOADH(J, T)                                                                               \
    T = z4;                                                                                         \
    if ((J) < NT && v < nht) {                                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) T[g] = sQd[(16 * (J) + lc) * LDQ + 16 * v + 4 * g + lq]; \
    }
    M_LOADH(0, th0) M_LOADH(1, th1) M_LOADH(2, th2) M_LOADH(3, th3)
#undef M_LOADH
    double mscale = sRed[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) mscale = fmax(mscale, sRed[k]);
    const double diag_thr = udbl(1e-4 * (mscale > 1.0 ? mscale : 1.0));
    double kx = (l < 16 && 16 * v + l < n) ? sQ[16 * v + l] : 0.0;       // lane l <-> row 16 v + l of the extra column
    __syncthreads();                    // the staged Qd has been read: the published pivot rows reuse its area
    STAMP(0);   // load

    // ---- rank-4 block pivots ---------------------------------------------------------------------------------
    // M_PUB(KB): gather of this wave's panel of pivot columns, and -- the owner of the pivot rows -- their raw values in
    // every live column tile, the raw pivot block and the rows' extra-column entries;  M_ELIM(KB): behind the barrier,
    // every wave factors the block and updates its row tile.
    bool fail = false;
#define M_PUBT(J, T, GP) if constexpr ((J) < NT) sVp[lq * VLD + 16 * (J) + lc] = T[GP];
#define M_PUBC(J, T, GP) if constexpr ((J) < MT) sVp[lq * VLD + 64 + 16 * (J) + lc] = T[GP];
#define M_PUB(KB, THJP)                                                                             \
    if ((KB) / 4 < NT && !fail && 4 * (KB) < n) {                                                   \
        constexpr int JP = (KB) / 4, GP = (KB) % 4, cq = 4 * GP, par = (KB) & 1;                    \
        double *const sVp = sV + par * 4 * VLD;                                                     \
        double *const sPp = sPr + par * 24;                                                         \
        if (v < nht) {                                                                              \
            const int kcol = lc - cq;                                                               \
            if (kcol >= 0 && kcol < 4) {                                                            \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                     \
                    double val = THJP[g];                                                           \
                    if (v == JP && g == GP) { sPp[lq * 4 + kcol] = val; if (lq == kcol) val -= 1.0; } \
                    sUv[(4 * g + lq) * 4 + kcol] = val;                                             \
                }                                                                                   \
            }                                                                                       \
            if (v == JP) {          /* the owner of the pivot rows */                               \
                if constexpr (JP <= 0) M_PUBT(0, th0, GP)                                           \
                if constexpr (JP <= 1) M_PUBT(1, th1, GP)                                           \
                if constexpr (JP <= 2) M_PUBT(2, th2, GP)                                           \
                M_PUBT(3, th3, GP)                                                                  \
                M_PUBC(0, tc0, GP) M_PUBC(1, tc1, GP) M_PUBC(2, tc2, GP) M_PUBC(3, tc3, GP)         \
                if (l >= cq && l < cq + 4) sPp[16 + l - cq] = kx;       /* their extra-column entries x_piv */ \
            }                                                                                       \
        }                                                                                           \
        __syncthreads();                                                                            \
        STAMP(1);                                                                                   \
    }
#define M_UPD(J, T) if constexpr ((J) < NT) { const double vr_ = sVp[lq * VLD + 16 * (J) + lc]; T = MFMA_NEGA(au, vr_, T); }
#define M_UPDC(J, T) if constexpr ((J) < MT) { const double vr_ = sVp[lq * VLD + 64 + 16 * (J) + lc]; T = MFMA_NEGA(au, vr_, T); }
#define M_ELIM(KB)                                                                                  \
    if ((KB) / 4 < NT && !fail && 4 * (KB) < n) {                                                   \
        constexpr int JP = (KB) / 4, par = (KB) & 1;                                                \
        const double *const sVp = sV + par * 4 * VLD;                                               \
        const double *const sPp = sPr + par * 24;                                                   \
        /* P = L U (unit lower L, no pivoting; uniform: every lane of every wave) */                \
        double pm[4][4];                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
            const d4 row = *reinterpret_cast<const d4 *>(sPp + i * 4);                              \
            pm[i][0] = row[0]; pm[i][1] = row[1]; pm[i][2] = row[2]; pm[i][3] = row[3];             \
        }                                                                                           \
        bool okp = true;                                                                            \
        double rd[4];                                                                               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
            okp = okp & (fabs(pm[s][s]) >= diag_thr);      /* (no short circuit: no branches in the factorization) */ \
            rd[s] = rcp64(pm[s][s]);                                                                \
            _Pragma("unroll") for (int i = s + 1; i < 4; ++i) {                                     \
                const double f = pm[i][s] * rd[s];                                                  \
                pm[i][s] = f;                                                                       \
                _Pragma("unroll") for (int j = s + 1; j < 4; ++j) pm[i][j] = fma(-f, pm[s][j], pm[i][j]); \
            }                                                                                       \
        }                                                                                           \
        if (!ubool(okp)) { fail = true; }                                                           \
        else if (v < nht) {                                                                         \
            /* column lq of P^-1 (L y = e_lq, U x = y): U' = U P^-1 straight in the A-operand layout -- lane (lc, lq) forms \
               U'[row lc][column lq]; pivot rows hold P - I, so theirs is I - P^-1 and the update turns them into P^-1 V */ \
            const double e0 = lq == 0 ? 1.0 : 0.0, e1 = lq == 1 ? 1.0 : 0.0, e2 = lq == 2 ? 1.0 : 0.0, e3 = lq == 3 ? 1.0 : 0.0; \
            const double y1 = fma(-pm[1][0], e0, e1);                                               \
            const double y2 = fma(-pm[2][1], y1, fma(-pm[2][0], e0, e2));                           \
            const double y3 = fma(-pm[3][2], y2, fma(-pm[3][1], y1, fma(-pm[3][0], e0, e3)));       \
            const double p3 = y3 * rd[3];                                                           \
            const double p2 = fma(-pm[2][3], p3, y2) * rd[2];                                       \
            const double p1 = fma(-pm[1][3], p3, fma(-pm[1][2], p2, y1)) * rd[1];                   \
            const double p0 = fma(-pm[0][3], p3, fma(-pm[0][2], p2, fma(-pm[0][1], p1, e0))) * rd[0]; \
            const d4 ur = *reinterpret_cast<const d4 *>(sUv + lc * 4);                              \
            const double au = fma(ur[3], p3, fma(ur[2], p2, fma(ur[1], p1, ur[0] * p0)));           \
            /* extra column: kx_l -= sum_q U'[l][q] x_piv[q] -- the four terms of row l sit in lanes l, l + 16, l + 32, l + 48 */ \
            kx -= xsum_rows(au * sPp[16 + lq]);                                                     \
            STAMP(2);                                                                               \
            if constexpr (JP <= 0) M_UPD(0, th0)                                                    \
            if constexpr (JP <= 1) M_UPD(1, th1)                                                    \
            if constexpr (JP <= 2) M_UPD(2, th2)                                                    \
            M_UPD(3, th3)                                                                           \
            M_UPDC(0, tc0) M_UPDC(1, tc1) M_UPDC(2, tc2) M_UPDC(3, tc3)                             \
            wave_sync();                                                                            \
            STAMP(3);                                                                               \
        }                                                                                           \
    }
    M_PUB(0, th0) M_ELIM(0) M_PUB(1, th0) M_ELIM(1) M_PUB(2, th0) M_ELIM(2) M_PUB(3, th0) M_ELIM(3)
    M_PUB(4, th1) M_ELIM(4) M_PUB(5, th1) M_ELIM(5) M_PUB(6, th1) M_ELIM(6) M_PUB(7, th1) M_ELIM(7)
    M_PUB(8, th2) M_ELIM(8) M_PUB(9, th2) M_ELIM(9) M_PUB(10, th2) M_ELIM(10) M_PUB(11, th2) M_ELIM(11)
    M_PUB(12, th3) M_ELIM(12) M_PUB(13, th3) M_ELIM(13) M_PUB(14, th3) M_ELIM(14) M_PUB(15, th3) M_ELIM(15)
#undef M_ELIM
#undef M_PUB
#undef M_PUBT
#undef M_PUBC
#undef M_UPD
#undef M_UPDC
    if (fail) { decline(); return; }

    l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
    // A operands of the S product (16 x 4 blocks of Ad, element (i = lc, k = lq)): all of them requested now, so that the
    // round trip hides behind the W~ hand-over (the H tiles are dead: their registers are free)
    const int arow = 16 * v + lc;
    double aop[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int r = 4 * kk + lq;                                  // k index: column of Ad, row of W~
        const bool valid = kk < 4 * NT && v < mct && r < n && arow < m;
        const double t_ = A_[valid ? (size_t)r * m + arow : 0];
        aop[kk] = valid ? t_ : 0.0;
    }
    // ---- W~ and h to LDS for the S product (the tiles keep W~ for the read-back) -----------------------------------
    __syncthreads();                                   // the last step's pivot rows have been read (sW reuses them)
    const int rot = mct == 4 ? lq : (mct == 3 ? (lq == 3 ? 0 : lq) : (mct == 2 ? (lq & 1) : 0));     // lq mod mct (compile-time mct)
    if (v < nht) {
#define M_WOUT(J, T)                                                                                \
    if constexpr ((J) < MT) {                                                                       \
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

    // ---- S = Ad W~ (row tile v), c = b - Ad h ------------------------------------------------------------------
    d4 s0 = z4, s1 = z4, s2 = z4, s3 = z4, sx = z4;
    if (v < mct) {
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
            if (kk < 4 * NT) {
                const int r = 4 * kk + lq;
                const double a_ = aop[kk];
                const double *wr = sW + r * m_pad + lc;
                s0 = MFMA(a_, wr[16 * j0], s0);
                if constexpr (MT > 1) s1 = MFMA(a_, wr[16 * j1], s1);
                if constexpr (MT > 2) s2 = MFMA(a_, wr[16 * j2], s2);
                if constexpr (MT > 3) s3 = MFMA(a_, wr[16 * j3], s3);
                const double hb = lc == 0 ? sH[r] : 0.0;
                sx = MFMA_NEGA(a_, hb, sx);
            }
        }
    }
    __syncthreads();                                   // W~ in LDS has been read: S goes through the same area
    // ---- S from row tiles to COLUMN tiles: wave v hands tile (v, J) to wave J (lane-contiguous blocks: no transposition
    // arithmetic, no bank conflicts); c goes to the leader
    double *const sX = sm + OFF_BIG;
    double *const colP = sm + OFF_COL;          // entering column, two buffers: row i at [i & 3][i >> 2], 16 doubles per i & 3
    if (v < mct) {
#define M_SOUT(J, T)                                                                                \
    if constexpr ((J) < MT) {                                                                       \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) sX[((v * MT + (J)) * 4 + g) * 64 + l] = T[g]; \
    }
        M_SOUT(0, s0) M_SOUT(1, s1) M_SOUT(2, s2) M_SOUT(3, s3)
#undef M_SOUT
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (lc == 0) colP[64 + 16 * v + 4 * g + lq] = sx[g];
    }
    __syncthreads();
    // the dictionary tiles of this wave -- column tile v, rows 16 I + 4 g + lq, column 16 v + lc at register 4 I + g --:
    // sixteen NAMED scalars (never an array or a vector: every dynamic select of one of them happens inside an asm dispatch
    // on a scalar index, so the compiler never copies or spills the set)
#define TD(j) td_##j
#define FOR_T(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define M_TLOAD(j) double TD(j) = ((j) < 4 * MT && v < mct) ? sX[((((j) >> 2) * MT + v) * 4 + ((j) & 3)) * 64 + l] : 0.0;
    FOR_T(M_TLOAD)
#undef M_TLOAD
    double xb = (v == 0 && actb) ? colP[64 + l] : 0.0;
    __syncthreads();                                   // S has been read: Ad for the post-check goes into the same area

    l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
    // Ad for the post-check: requested now (coalesced), parked in LDS
    const int LDA = m | 1;
    double *const sAd = sm + OFF_BIG;
    {
        constexpr int TPB_ = 64 * NW;
        for (int e0 = tid; e0 < m * n; e0 += 4 * TPB_) {
            double vv[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { const int e = e0 + q4 * TPB_; vv[q4] = e < m * n ? A_[e] : 0.0; }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int e = e0 + q4 * TPB_;
                const int j = e / m, i = e - j * m;
                if (e < m * n) sAd[j * LDA + i] = vv[q4];
            }
        }
    }

    // ================= Stage B: Lemke on the Schur dictionary =================
    // pair k (k < m) <-> item row n + k:  p_k = (S lambda + c)_k in [l_k, u_k],  d_k = lambda_k.
    // ids: p_k -> k, d_k -> 64 + k, artificial -> 128; column index 64 = the extra (covering) column.
    // Wave 0 is the LEADER: it alone keeps the bookkeeping (row vectors lane <-> row, column vectors lane <-> column, the extra
    // column as the row vector tcol) and runs the ratio test; the other waves only hold their tiles.  Per pivot:
    //   the owner of the entering column publishes it (the 4 lanes that hold it; the leader for the extra column) -- barrier A --
    //   the leader reads it lane <-> row, picks the row, does the bookkeeping and posts {what, row, 1 / pivot, next column} --
    //   barrier B -- every wave runs the rank-1 exchange on its tiles as in the 32-class kernel: its piece of the pivot row
    //   through a wave-private LDS vector (no barrier), its 4 MT column entries from the published column, 4 MT v_fma_f64, the
    //   pivot column rescaled on its 4 owner lanes and the pivot row rewritten on its 16 owner lanes under EXEC masks.
    constexpr int NBP = 64, XC = 64, VTH = 128;
    constexpr int CODE_PIVOT = 0, CODE_FLIP = 1, CODE_STOP = 2;
    double *const sLo = sm + OFF_U, *const sHi = sm + OFF_U + 64;      // fixed pair bounds (looked up by a uniform index)
    double *const rowbuf = sm + OFF_U + 128 + 16 * v;                  // this wave's piece of the pivot row             [16]
    double *const sDecD = sm + OFF_U + 192;                            // the leader's decision: 1 / pivot                [1]
    int *const sDecI = reinterpret_cast<int *>(sm + OFF_U + 194);      // ... what, row, next column                      [4]
    const bool lead = v == 0;
    double lo = -QINF, hi = QINF;
    if (lead && actb) { lo = a.nd.l[(size_t)b * m + l]; hi = a.nd.u[(size_t)b * m + l]; }
    if (lead) { sLo[l] = lo; sHi[l] = hi; }
    int satv = 0;
    int rowvar = actb ? l : -1, colvar = actb ? NBP + l : -1;
    int cvx = VTH;
    double nbx = 0.0, nbval = 0.0, tcol = 0.0;
    int pivots = n;                       // the crash brought n free variables in (Stage A)
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;
    int status = QPN_FAILURE;
    int c = XC, par = 0;
    bool sneg = true;
    double self_lim = 0.0, elo = 0.0, ehi = QINF;
    const double slack = 1e-10, ptol = a.piv_tol;
    auto col_of = [&](int var) -> int {
        const int cc = wave_first(colvar == var);
        return cc >= 0 ? cc : (cvx == var ? XC : -1);
    };
    if (lead) {
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
            nbx = theta0;
            self_lim = theta0;
            status = QPN_MAX_ITERS;
        }
        if (l == 0) sDecI[0] = status == QPN_MAX_ITERS ? CODE_PIVOT : CODE_STOP;
    }
    __syncthreads();
    STAMP(4);   // W~ hand-over, S product, tile hand-over, Ad staging
    int code = uni(sDecI[0]);
    // sixteen-way scalar dispatch on a wave-uniform index (0..15; anything else: nothing): leaf k names register k statically
#define T_OPS_RW [t0] "+v"(TD(0)), [t1] "+v"(TD(1)), [t2] "+v"(TD(2)), [t3] "+v"(TD(3)), [t4] "+v"(TD(4)), [t5] "+v"(TD(5)),       \
                 [t6] "+v"(TD(6)), [t7] "+v"(TD(7)), [t8] "+v"(TD(8)), [t9] "+v"(TD(9)), [t10] "+v"(TD(10)), [t11] "+v"(TD(11)),   \
                 [t12] "+v"(TD(12)), [t13] "+v"(TD(13)), [t14] "+v"(TD(14)), [t15] "+v"(TD(15))
#define T_OPS_R [t0] "v"(TD(0)), [t1] "v"(TD(1)), [t2] "v"(TD(2)), [t3] "v"(TD(3)), [t4] "v"(TD(4)), [t5] "v"(TD(5)),             \
                [t6] "v"(TD(6)), [t7] "v"(TD(7)), [t8] "v"(TD(8)), [t9] "v"(TD(9)), [t10] "v"(TD(10)), [t11] "v"(TD(11)),         \
                [t12] "v"(TD(12)), [t13] "v"(TD(13)), [t14] "v"(TD(14)), [t15] "v"(TD(15))
#define DISPATCH16(P, L0, L1, L2, L3, L4, L5, L6, L7, L8, L9, L10, L11, L12, L13, L14, L15)                                        \
    /* (numeric local labels, all referenced forwards: the compiler may duplicate an asm statement -- loop peeling --, and named  \
        labels would then be defined twice; P only documents the call site) */                                                     \
    "s_cmp_gt_u32 %[cs], 7\n\ts_cbranch_scc1 20f\n\t"                                                                             \
    "s_cmp_gt_u32 %[cs], 3\n\ts_cbranch_scc1 4f\n\t"                                                                              \
    "s_cmp_gt_u32 %[cs], 1\n\ts_cbranch_scc1 2f\n\t"                                                                              \
    "s_cmp_eq_u32 %[cs], 0\n\ts_cbranch_scc0 1f\n\t"                                                                              \
    L0 "\n\ts_branch 30f\n"                                                                                                        \
    "1:\n\t" L1 "\n\ts_branch 30f\n"                                                                                              \
    "2:\n\ts_cmp_eq_u32 %[cs], 2\n\ts_cbranch_scc0 3f\n\t" L2 "\n\ts_branch 30f\n"                                              \
    "3:\n\t" L3 "\n\ts_branch 30f\n"                                                                                              \
    "4:\n\ts_cmp_gt_u32 %[cs], 5\n\ts_cbranch_scc1 6f\n\t"                                                                       \
    "s_cmp_eq_u32 %[cs], 4\n\ts_cbranch_scc0 5f\n\t" L4 "\n\ts_branch 30f\n"                                                    \
    "5:\n\t" L5 "\n\ts_branch 30f\n"                                                                                              \
    "6:\n\ts_cmp_eq_u32 %[cs], 6\n\ts_cbranch_scc0 7f\n\t" L6 "\n\ts_branch 30f\n"                                              \
    "7:\n\t" L7 "\n\ts_branch 30f\n"                                                                                              \
    "20:\n\ts_cmp_gt_u32 %[cs], 15\n\ts_cbranch_scc1 30f\n\t"                                                                    \
    "s_cmp_gt_u32 %[cs], 11\n\ts_cbranch_scc1 12f\n\t"                                                                            \
    "s_cmp_gt_u32 %[cs], 9\n\ts_cbranch_scc1 10f\n\t"                                                                             \
    "s_cmp_eq_u32 %[cs], 8\n\ts_cbranch_scc0 9f\n\t" L8 "\n\ts_branch 30f\n"                                                    \
    "9:\n\t" L9 "\n\ts_branch 30f\n"                                                                                              \
    "10:\n\ts_cmp_eq_u32 %[cs], 10\n\ts_cbranch_scc0 11f\n\t" L10 "\n\ts_branch 30f\n"                                         \
    "11:\n\t" L11 "\n\ts_branch 30f\n"                                                                                            \
    "12:\n\ts_cmp_gt_u32 %[cs], 13\n\ts_cbranch_scc1 14f\n\t"                                                                    \
    "s_cmp_eq_u32 %[cs], 12\n\ts_cbranch_scc0 13f\n\t" L12 "\n\ts_branch 30f\n"                                                 \
    "13:\n\t" L13 "\n\ts_branch 30f\n"                                                                                            \
    "14:\n\ts_cmp_eq_u32 %[cs], 14\n\ts_cbranch_scc0 15f\n\t" L14 "\n\ts_branch 30f\n"                                         \
    "15:\n\t" L15 "\n"                                                                                                            \
    "30:\n\t"
    while (code != CODE_STOP) {
        c = uni(c);
        // ---- the entering column, published by the lanes that hold it: row i goes to [i & 3][i >> 2]
        {
            double *const cp = colP + par * 64;
            if (c == XC) { if (lead) cp[(l & 3) * 16 + (l >> 2)] = tcol; }
            else if (v == (c >> 4) && lc == (c & 15)) {
                d4 *const dst = reinterpret_cast<d4 *>(cp + lq * 16);
                dst[0] = d4{TD(0), TD(1), TD(2), TD(3)};
                if constexpr (MT > 1) dst[1] = d4{TD(4), TD(5), TD(6), TD(7)};
                if constexpr (MT > 2) dst[2] = d4{TD(8), TD(9), TD(10), TD(11)};
                if constexpr (MT > 3) dst[3] = d4{TD(12), TD(13), TD(14), TD(15)};
            }
        }
        __syncthreads();                                // barrier A
        STAMP(5);   // entering column through LDS + barrier A
        if (lead) {
            // ================= the leader's turn: ratio test, bookkeeping, decision =================
            const double cm = actb ? colP[par * 64 + (l & 3) * 16 + (l >> 2)] : 0.0;      // (rows beyond m: whatever LDS holds there)
            // ratio test (two-pass Harris with 1e-10 slack; largest pivot among ties, the artificial first)
            const double gdir = __hiloint2double(__double2hiint(cm) ^ (sneg ? (int)0x80000000 : 0), __double2loint(cm));
            const double rc = rcp64(gdir);
            const bool gneg = __double2hiint(gdir) < 0;
            const double tb = gneg ? lo : hi;                   // the bound the row's basic variable moves towards
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wbitwise-instead-of-logical"
            const bool cnd = (actb & (fabs(gdir) > ptol)) & (fabs(tb) < QINF);
#pragma clang diagnostic pop
            const double arc = fabs(rc);
            const double dd = cnd ? (tb - xb) * rc : QINF;
            const double d1 = fma(slack, arc, dd);
            const double dmax = wave_min64_with_limit_f64(d1, self_lim);
            int dcode = CODE_PIVOT, r = 0, cnext = XC;
            double inv = 0.0;
            if (pivots >= max_piv) dcode = CODE_STOP;          // status stays MAX_ITERS
            else if (uni(__double2hiint(dmax)) == 0x7ff00000) { status = QPN_RAY_TERM; dcode = CODE_STOP; }
            else {
                const unsigned long long bal = qpn_ballot(dd <= dmax);
                const int ve = (c == XC) ? cvx : readlane_i32(colvar, c);      // the entering variable
                const double eloW = elo, ehiW = ehi;                            // ... and the interval it lives in once basic
                double delta, vx = 0.0, tc0 = tcol, enter_val = 0.0, nbW = 0.0;
                int rW = -1, cW = -1, kW = -1, auW = 0, vlW = ve;
                const bool newx = c == XC;
                if (bal == 0ull) {
                    // the entering variable reaches its own opposite bound first: no basis change
                    dcode = CODE_FLIP;
                    delta = sneg ? -self_lim : self_lim;
                    cW = c;
                    if (ve == VTH) { nbW = 0.0; status = QPN_SUCCESS; dcode = CODE_STOP; }
                    else {
                        const int k = ve;
                        const int au = sneg ? 0 : 1;
                        nbW = udbl(au ? sHi[k] : sLo[k]);
                        kW = k; auW = au;
                        pivots++;
                        sneg = au != 0;
                        self_lim = QINF;
                        if (au) { elo = -QINF; ehi = 0.0; } else { elo = 0.0; ehi = QINF; }
                        // (the column of d_k: the bookkeeping vectors do not change in a flip, so the search can run here)
                        cnext = col_of(NBP + k);
                        if (cnext < 0) { status = QPN_FAILURE; dcode = CODE_STOP; }
                    }
                } else {
                    if (__popcll(bal) == 1) r = __ffsll((long long)bal) - 1;
                    else {
                        const bool cand = dd <= dmax;
                        double ag = cand ? fabs(gdir) : -1.0;
                        if (cand && rowvar == VTH) ag = QINF;
                        const double bestg = wave_max_f64(ag);
                        r = wave_first(cand && ag == bestg);
                    }
                    r = uni(r);
                    double step = readlane_f64(dd, r);
                    if (step < 0.0) step = 0.0;
                    const double leave_val = readlane_f64(tb, r);
                    const double rcr = readlane_f64(rc, r);
                    inv = sneg ? -rcr : rcr;                    // 1 / T[r][c]
                    delta = sneg ? -step : step;
                    const int vl = readlane_i32(rowvar, r);
                    enter_val = (newx ? nbx : readlane_f64(nbval, c)) + delta;
                    // the extra column (scalar statement: prow = T[r][.] * inv; T[i][.] = fma(-cm_i, prow, T[i][.]); row r:
                    // -prow; the pivot column's own entries start from 0 and its slot in the row carries -inv)
                    if (newx) { vx = -inv; tc0 = 0.0; } else vx = readlane_f64(tcol, r) * inv;
                    rW = r; cW = c; vlW = vl; nbW = leave_val;
                    pivots++;
                    if (vl == VTH) { status = QPN_SUCCESS; dcode = CODE_STOP; }
                    else {
                        int vn;
                        const int k = vl < NBP ? vl : vl - NBP;
                        const double lk0 = udbl(sLo[k]), hk0 = udbl(sHi[k]);
                        const int cls = (lk0 == -QINF && hk0 == QINF) ? 2 : 0;
                        if (vl < NBP) {
                            // the bounded variable p_k left at a bound -- the upper one iff row r was a `hi` ratio --: d_k enters
                            const int au = uni(__double2hiint(rcr)) >= 0 ? 1 : 0;
                            kW = k; auW = au;
                            vn = NBP + k;
                            sneg = au != 0;
                            self_lim = QINF;
                            if (cls == 2) { elo = 0.0; ehi = 0.0; }
                            else if (au) { elo = -QINF; ehi = 0.0; }
                            else { elo = 0.0; ehi = QINF; }
                        } else {
                            // the multiplier d_k left at 0: p_k enters, moving off the bound it rests at
                            const int au = readlane_i32(satv, k);
                            vn = k;
                            sneg = au != 0;
                            self_lim = udbl(hk0 - lk0);         // +inf for a free pair
                            if (cls == 2) sneg = false;
                            elo = lk0; ehi = hk0;
                        }
                        // (colvar / cvx still hold the entering id at column c here -- the write-back is below --, which is
                        //  never vn's; the id that lands there, vl, is never its own complement vn either)
                        cnext = (vn == ve) ? -1 : col_of(vn);
                        if (cnext < 0) { status = QPN_FAILURE; dcode = CODE_STOP; }
                    }
                }
                // values, the extra column, bookkeeping write-backs (one definition of every vector per iteration)
                const bool isr = l == rW;
                {
                    const double xbn = fma(delta, cm, xb);
                    const double tcn = fma(-cm, vx, tc0);
                    xb = isr ? enter_val : xbn;
                    tcol = isr ? -vx : tcn;
                }
                if (isr) { rowvar = ve; lo = eloW; hi = ehiW; }
                {
                    const bool isc = !newx && l == cW;
                    if (isc) { colvar = vlW; nbval = nbW; }
                    if (newx) { cvx = vlW; nbx = nbW; }
                    if (l == kW) satv = auW;
                }
            }
            if (l == 0) { sDecI[0] = dcode; sDecI[1] = r; sDecI[2] = cnext; sDecD[0] = inv; }
        }
        __syncthreads();                                // barrier B
        STAMP(6);   // the leader's turn + barrier B
        code = uni(sDecI[0]);
        const int r = uni(sDecI[1]), cnext = uni(sDecI[2]);
        const double inv = udbl(sDecD[0]);
        if (code == CODE_STOP) break;
        // ---- the exchange (a flip runs it with empty lane masks and a zero row: a no-op, so the tile registers have one
        // definition per iteration)
        {
            const bool piv = code == CODE_PIVOT;
            const int rq = r & 3, rsel = piv ? (r >> 2) : 16;
            const bool own = piv && c != XC && v == (c >> 4);
            const unsigned long long mrow = piv ? 0xFFFFull << (16 * rq) : 0ull;
            const unsigned long long mcol = own ? 0x0001000100010001ull << (c & 15) : 0ull;
            // this wave's piece of the pivot row: the 16 lanes that hold it pick the register under the scalar dispatch
            {
                double *const dst = rowbuf + lc;
                asm volatile("s_mov_b64 exec, %[mr]\n\t"
                             DISPATCH16("Lwgr", "ds_write_b64 %[ad], %[t0]", "ds_write_b64 %[ad], %[t1]", "ds_write_b64 %[ad], %[t2]",
                                        "ds_write_b64 %[ad], %[t3]", "ds_write_b64 %[ad], %[t4]", "ds_write_b64 %[ad], %[t5]",
                                        "ds_write_b64 %[ad], %[t6]", "ds_write_b64 %[ad], %[t7]", "ds_write_b64 %[ad], %[t8]",
                                        "ds_write_b64 %[ad], %[t9]", "ds_write_b64 %[ad], %[t10]", "ds_write_b64 %[ad], %[t11]",
                                        "ds_write_b64 %[ad], %[t12]", "ds_write_b64 %[ad], %[t13]", "ds_write_b64 %[ad], %[t14]",
                                        "ds_write_b64 %[ad], %[t15]")
                             "s_mov_b64 exec, -1"
                             :
                             : T_OPS_R, [cs] "s"(uni(rsel)), [mr] "s"(mrow),
                               [ad] "v"((unsigned)(size_t)(__attribute__((address_space(3))) double *)dst)
                             : "scc", "memory");
                // the entry of the pivot column itself is replaced by -1, so that row * inv carries -inv there (what the
                // exchange needs) without any select
                if (own && l == 16 * rq + (c & 15)) rowbuf[c & 15] = -1.0;
            }
            wave_sync();
            const double pv = piv ? rowbuf[lc] * inv : 0.0;
            // this lane's column entries: rows lq + 4 k, k = 4 I + g
            const d4 *const up = reinterpret_cast<const d4 *>(colP + par * 64 + lq * 16);
            const d4 u0 = up[0], u1 = MT > 1 ? up[1] : z4, u2 = MT > 2 ? up[2] : z4, u3 = MT > 3 ? up[3] : z4;
            const double inv_s = udbl(inv);
#define M_XCHG8(A0, A1, A2, A3, A4, A5, A6, A7, UA, UB)                                                                                                    \
            asm volatile("v_fma_f64 %[a0], -%[u0], %[pv], %[a0]\n\tv_fma_f64 %[a1], -%[u1], %[pv], %[a1]\n\t"                      \
                         "v_fma_f64 %[a2], -%[u2], %[pv], %[a2]\n\tv_fma_f64 %[a3], -%[u3], %[pv], %[a3]\n\t"                      \
                         "v_fma_f64 %[a4], -%[u4], %[pv], %[a4]\n\tv_fma_f64 %[a5], -%[u5], %[pv], %[a5]\n\t"                      \
                         "v_fma_f64 %[a6], -%[u6], %[pv], %[a6]\n\tv_fma_f64 %[a7], -%[u7], %[pv], %[a7]\n\t"                      \
                         /* column c: T[i][c] = u_i * inv on the 4 lanes that own it */                                            \
                         "s_mov_b64 exec, %[mc]\n\ts_cbranch_execz 31f\n\t"                                                  \
                         "v_mul_f64 %[a0], %[u0], %[iv]\n\tv_mul_f64 %[a1], %[u1], %[iv]\n\t"                                      \
                         "v_mul_f64 %[a2], %[u2], %[iv]\n\tv_mul_f64 %[a3], %[u3], %[iv]\n\t"                                      \
                         "v_mul_f64 %[a4], %[u4], %[iv]\n\tv_mul_f64 %[a5], %[u5], %[iv]\n\t"                                      \
                         "v_mul_f64 %[a6], %[u6], %[iv]\n\tv_mul_f64 %[a7], %[u7], %[iv]\n"                                        \
                         "31:\n\ts_mov_b64 exec, -1"                                                                        \
                         : [a0] "+v"(TD(A0)), [a1] "+v"(TD(A1)), [a2] "+v"(TD(A2)), [a3] "+v"(TD(A3)),                             \
                           [a4] "+v"(TD(A4)), [a5] "+v"(TD(A5)), [a6] "+v"(TD(A6)), [a7] "+v"(TD(A7))                              \
                         : [u0] "v"(UA[0]), [u1] "v"(UA[1]), [u2] "v"(UA[2]), [u3] "v"(UA[3]), [u4] "v"(UB[0]), [u5] "v"(UB[1]),  \
                           [u6] "v"(UB[2]), [u7] "v"(UB[3]), [pv] "v"(pv), [iv] "s"(inv_s), [mc] "s"(mcol));
            M_XCHG8(0, 1, 2, 3, 4, 5, 6, 7, u0, u1)
            if constexpr (MT > 2) { M_XCHG8(8, 9, 10, 11, 12, 13, 14, 15, u2, u3) }
#undef M_XCHG8
            // row r: T[r][j] = -pv_j on the 16 lanes that own it (pv carries -inv at column c)
            asm volatile("s_mov_b64 exec, %[mr]\n\t"
                         DISPATCH16("Lwgf", "v_mul_f64 %[t0], %[pv], -1.0", "v_mul_f64 %[t1], %[pv], -1.0", "v_mul_f64 %[t2], %[pv], -1.0",
                                    "v_mul_f64 %[t3], %[pv], -1.0", "v_mul_f64 %[t4], %[pv], -1.0", "v_mul_f64 %[t5], %[pv], -1.0",
                                    "v_mul_f64 %[t6], %[pv], -1.0", "v_mul_f64 %[t7], %[pv], -1.0", "v_mul_f64 %[t8], %[pv], -1.0",
                                    "v_mul_f64 %[t9], %[pv], -1.0", "v_mul_f64 %[t10], %[pv], -1.0", "v_mul_f64 %[t11], %[pv], -1.0",
                                    "v_mul_f64 %[t12], %[pv], -1.0", "v_mul_f64 %[t13], %[pv], -1.0", "v_mul_f64 %[t14], %[pv], -1.0",
                                    "v_mul_f64 %[t15], %[pv], -1.0")
                         "s_mov_b64 exec, -1"
                         : T_OPS_RW
                         : [cs] "s"(uni(rsel)), [mr] "s"(mrow), [pv] "v"(pv)
                         : "scc");
        }
        c = cnext;
        par ^= 1;
        STAMP(7);   // the exchange
    }
#undef DISPATCH16
#undef T_OPS_R
#undef T_OPS_RW
#undef FOR_T
#undef TD

    // ---- read back: lambda_k, then x = W~ lambda - h -------------------------------------------------------------
    // (everything below derives its lane coordinates and kernel arguments afresh: nothing of that stays in registers across
    //  the pivot loop)
    l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
    typedef const AviBatchArgs __attribute__((address_space(4))) *kargs_t;
    kargs_t kp = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    __syncthreads();                                   // every wave is out of the loop (sval / sz lie over Stage-A scratch only,
                                                       // but the Ad stores above must have landed before the post-check)
    if (v == 0) {
        if (l < m) { sval[rowvar] = xb; sval[colvar] = nbval; }
        else sval[NBP + l] = 0.0;                                       // multipliers of the padded columns of W~ (finite: 0 x 0)
        if (l == 0) sval[cvx] = nbx;
    }
    __syncthreads();
    if (v < nht) {
        // (W~ lambda)_row = sum over the 16 lanes of a DPP row of this lane's partial over its column of each tile, for the
        // 4 rows (g) a lane holds.  Folded butterfly: at each of the first two stages a lane gives half of its values to its
        // partner and adds the partner's other half (4 -> 2 -> 1 values; lane bits 0, 1 of lc then name g), bits 2, 3 are
        // summed by two full stages (xor 4, xor 8: ds_swizzle, the LDS crossbar)
        double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
        l0 = sval[NBP + lc];
        if constexpr (MT > 1) l1 = sval[NBP + 16 + lc];
        if constexpr (MT > 2) l2 = sval[NBP + 32 + lc];
        if constexpr (MT > 3) l3 = sval[NBP + 48 + lc];
        const double p0 = fma(tc3[0], l3, fma(tc2[0], l2, fma(tc1[0], l1, tc0[0] * l0)));
        const double p1 = fma(tc3[1], l3, fma(tc2[1], l2, fma(tc1[1], l1, tc0[1] * l0)));
        const double p2 = fma(tc3[2], l3, fma(tc2[2], l2, fma(tc1[2], l1, tc0[2] * l0)));
        const double p3 = fma(tc3[3], l3, fma(tc2[3], l2, fma(tc1[3], l1, tc0[3] * l0)));
        const bool b0 = (lc & 1) != 0, b1 = (lc & 2) != 0;
        const double q0 = (b0 ? p1 : p0) + dpp_f64<0xB1>(b0 ? p0 : p1), q1 = (b0 ? p3 : p2) + dpp_f64<0xB1>(b0 ? p2 : p3);
        double t0 = (b1 ? q1 : q0) + dpp_f64<0x4E>(b1 ? q0 : q1);       // g = b0 + 2 b1, summed over lc bits 0, 1
        t0 += __shfl_xor(t0, 4, 64);
        t0 += __shfl_xor(t0, 8, 64);
        if (lc < 4) {
            const int row = 16 * v + 4 * lc + lq;                       // g = lc for lc < 4
            sz[row] = t0 - sH[row];
        }
    }
    __syncthreads();
    // item order: rows < n are x (already in sz), rows n.. are lambda
    if (tid < m) sz[n + tid] = sval[NBP + tid];
    __syncthreads();

    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -------------------------------------
    const double *const Qe_ = kp->nd.Qd + (size_t)b * n * n;
    const double *const le_ = kp->nd.l, *const ue_ = kp->nd.u;
    double *const ze_ = kp->z, *const xe_ = kp->x, *const re_ = kp->resid;
    uint8_t *const ae_ = kp->active;
    int32_t *const pe_ = kp->pivots;
    int bad = 0;
    double nres = 0.0;
    if (tid < N) {
        const int k = tid;
        const int gk = k >= n;
        double rk = sQ[k];
        // row k of [[Qd, -Ad'],[Ad, 0]] times z, columns ascending (finite blocks: a zero z_j contributes exactly nothing)
        // (tried: two lanes per row -- the workgroup has twice as many threads as the item has rows --, all of a lane's entries of
        //  Qd requested at once, two accumulators, v_permlane32_swap to add the halves: 1 % slower at 4 000 nodes and at 256)
        if (!gk) {
            int j = 0;
            for (; j + 8 <= n; j += 8) {
                double mv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) mv[q8] = Qe_[(size_t)(j + q8) * n + k];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) rk = fma(mv[q8], sz[j + q8], rk);
            }
            for (; j < n; ++j) rk = fma(Qe_[(size_t)j * n + k], sz[j], rk);
            for (int i = 0; i < m; ++i) rk = fma(-sAd[k * LDA + i], sz[n + i], rk);
        } else {
            const int r = k - n;
            for (int j = 0; j < n; ++j) rk = fma(sAd[j * LDA + r], sz[j], rk);
        }
        const double zk = sz[k];
        const double lk = gk ? le_[(size_t)b * m + (k - n)] : -QINF, uk = gk ? ue_[(size_t)b * m + (k - n)] : QINF;
        const double p = gk ? rk : zk, d = gk ? zk : rk;
        const double tol = kp->check_tol;
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
        const double ct = kp->comp_tol;
        auto approx = [&](double x, double y) { return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= ct); };
        if (!approx(lk, uk)) {
            if (approx(p, lk) && d >= -ct) mask |= 1u;
            if (lk - ct <= p && p <= uk + ct && fabs(d) <= ct) mask |= 2u;
            if (approx(p, uk) && d <= ct) mask |= 4u;
        } else mask = 8u;
        if (gk) mask <<= 4;
        ze_[(size_t)b * N + k] = zk;
        if (xe_ && !gk) {                                                       // primal block -> the caller's iterate
            const size_t xo = (size_t)b * (size_t)kp->stride_x + k;
            xe_[xo] = zk;
            for (int q = 0; q < kp->n_mirror; ++q) kp->mirror[q][xo] = zk;          // ... and its replicas on the peer GPUs
        }
        if (ae_) ae_[(size_t)b * N + k] = (uint8_t)mask;
    }
    const int badt = __syncthreads_count(bad > 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(nres, off, 64); nres = o > nres ? o : nres; }
    if (l == 0) sRed[v] = nres;
    __syncthreads();
    if (tid == 0) {
        if (badt > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
        kp->status[b] = status;
        double rs = sRed[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) rs = sRed[k] > rs ? sRed[k] : rs;
        if (re_) re_[b] = rs;
        if (pe_) pe_[b] = pivots;
        if (int32_t *const sk = kp->sched_key) { const int k0 = sk[b]; sk[b] = k0 > 0 ? k0 - (k0 >> 5) + pivots : 32 * pivots; }      // smoothed pivot count
    }
    STAMP(0);   // (diagnostic builds: read-back + post-check are added to the load slot)
#ifdef QPN_STAMPS
    if (tid == 0 && a.stamps) {
        for (int k = 0; k < 8; ++k) a.stamps[(size_t)b * 8 + k] = stamp_acc[k];
    }
    // where the waves of this workgroup ran (tools/wg_simd_probe.py): SIMD of wave v in bits 48 + 2 v of slot 0, the CU's id in bits 56..
    __syncthreads();
    if (l == 0 && a.stamps) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned long long bits = (unsigned long long)((hw >> 4) & 3u) << (48 + 2 * v);
        if (v == 0) bits |= (unsigned long long)((hw >> 8) & 0xFu) << 56;
        atomicOr(&a.stamps[(size_t)b * 8 + 0], bits);
        if (v == 0) {
            // ... and the whole identity of the leader's slot in bits 48.. of slot 1: HW_ID[15:0] (wave slot, SIMD, pipe, CU, SH, SE) and XCC_ID
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            atomicOr(&a.stamps[(size_t)b * 8 + 1], ((unsigned long long)(hw & 0xFFFFu) << 48) | ((unsigned long long)(xcc & 0xFu) << 44));
        }
    }
#endif
}

} // namespace

bool qpn_schur_wg_shape(int n, int m)
{
    return (n > 32 || m > 32) && n >= 1 && n <= 64 && m >= 1 && m <= 64;
}

// One launch: every node of the batch solved, checked and written back by its own workgroup.  Nodes the kernel declines
// keep status -1.
hipError_t qpn_launch_schur_wg_nodes(const AviBatchArgs &a, hipStream_t stream)
{
    const int n = a.nd.n, m = a.nd.m, batch = a.batch;
    if (batch <= 0) return hipSuccess;
    // size classes: both tile counts rounded up to 3 or 4 (a dimension of 32 or less rides in the 48-class of the other)
    const int NT = n <= 48 ? 3 : 4, MT = m <= 48 ? 3 : 4;
    const size_t lds = (size_t)wg_lds_doubles(n, m, 16 * NT, 16 * MT) * sizeof(double);
    const dim3 grid((unsigned)batch);
    // (max(n, m) <= 48 is the one-wavefront kernel's class, qpn_avi_schur48.hip: the dispatcher never sends it here)
    if (NT == 3 && MT == 3) return hipErrorInvalidValue;
    if (NT == 3) hipLaunchKernelGGL((schur_wg_nodes<3, 4>), grid, dim3(256), lds, stream, a);
    else if (MT == 3) hipLaunchKernelGGL((schur_wg_nodes<4, 3>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((schur_wg_nodes<4, 4>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}
