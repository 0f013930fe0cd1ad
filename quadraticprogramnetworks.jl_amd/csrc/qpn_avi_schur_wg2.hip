// qpn_avi_schur_wg2.hip -- the fused one-workgroup-per-node kernel for LARGER node records (n, m <= 128, one of them > 64),
// gfx950: the layout of qpn_avi_schur_wg.hip with TWO wavefronts per row tile.
//
// Round 2 sent these nodes down the route built for BASELINE config 5 (assembled M in HBM, panels of a row-major top half
// walked through LDS: 0.8 M solves/s at n = m = 96 against 6 M/s at 64).  Here a node is ONE workgroup of 2 NR wavefronts,
// NR = ceil(max(n, m) / 16) in 5 .. 8, and -- as in the 33..64 class -- nothing but the records is read and nothing but the
// outputs is written: KKT assembly (src/avi.jl:205-251, :305-377), crash, Lemke, read-back, post-check
// (src/avi.jl:71-76, :148-156), active sets (src/avi_solutions.jl:511-562), primal write-back (src/avi.jl:440-443).
//
//   Waves 0 .. NR-1 ("H waves"): wave I holds row tile I of H (16 rows x NR tiles: <= 64 VGPRs), the extra column g, and,
//     after the crash, first a row tile of S (the S product) and then COLUMN tile I of the Schur dictionary (Stage B).
//   Waves NR .. 2NR-1 ("C waves"): wave NR + I holds row tile I of C~ = +Ad' (16 rows x NR tiles), which the crash turns into
//     W~ = H^-1 Ad'; they keep it in registers through Stage B for the read-back x = W~ lambda - h.
//   Stage A: per rank-4 block pivot the two owners of the pivot rows publish them raw (B operands), the H owner also the
//     raw 4 x 4 pivot block, every H wave its 16 x 4 panel of pivot columns: ONE workgroup barrier; then EVERY wave factors
//     the pivot block itself (uniform, in registers) and forms its entry of U' = U P^-1 in the A-operand layout: one MFMA
//     (NEG on A) per live tile.  Qd reaches the H tiles through LDS in 16-column panels (coalesced whole-column loads, two
//     panels in flight), Ad reaches the C tiles directly.
//   S = Ad W~ and c = b - Ad h: W~ crosses LDS once, H wave I computes row tile I of S (A operands streamed from the
//     records); the tiles change hands through LDS (row tiles -> column tiles).
//   Stage B: as in qpn_avi_schur_wg.hip -- the H waves hold the dictionary in column tiles, wave 0 is the leader (ratio
//     test over up to 128 rows: two rows per lane; bookkeeping; decision), two barriers per pivot, the exchange of a wave is
//     4 NR v_fma_f64 plus lane-masked fix-ups, its piece of the pivot row through a wave-private LDS vector.
// Declined nodes (a block pivot below the threshold, an equality row) keep status -1 and take the general path in gated
// launches.  One workgroup per CU (the W~ hand-over alone is up to 128 KB of LDS).
#include "qpn_internal.h"

#define QINF __builtin_huge_val()

namespace {

constexpr int VLD = 272;                // row stride of the published pivot rows (256 columns; == 16 mod 32)
typedef double d4 __attribute__((ext_vector_type(4)));
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
// min over all 64 lanes of v and the wave-uniform `lim`, returned wave-uniform (see qpn_avi_schur_wg.hip)
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
__device__ __forceinline__ int lane_id_fresh()
{
    int x = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(x));
    return x;
}
// v + (v of lane ^ 16) + (v of lane ^ 32) + (v of lane ^ 48): gfx950's row / half swaps (VALU, no LDS trip)
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

// LDS map (doubles; ~151 KB at n = m = 128: one workgroup per CU).  The big area is, in turn: two Qd panels [2][16][n_pad + 2], the published pivot rows [2][4][VLD], W~
// for the S product [n_pad][m_pad], the S tiles changing hands [NR][NR][4][64], Ad for the post-check [n][m | 1].
constexpr int OFF_Q = 0;                // q = [g ; b] in item order                              [256]
constexpr int OFF_U = 256;              // Stage A: per row tile its 16 x 4 panel of pivot columns, two buffers (the H wave that
                                        //   writes a panel and the C wave that also reads it run apart)  [2][8][64];
                                        //   read-back: values by id [264], z [256]
constexpr int OFF_PR = 1280;            // Stage A: raw pivot block + x_piv, two buffers          [2][24]
constexpr int OFF_RED = 1328;           // block reduction                                        [16]
constexpr int OFF_H = 1344;             // h (the eliminated extra column)                        [128]
constexpr int OFF_COL = 1472;           // Stage B: the pivot column for the exchange, two buffers, and the next column as it
                                        //   stands; row i at [i & 3][i >> 2], 34 doubles per i & 3                    [3][136]
constexpr int OFF_B = 1880;             // Stage B, leader's tables: fixed pair bounds sLo, sHi [128 each], the rows' current
                                        //   intervals sRowLo, sRowHi [128 each], nonbasic values by column [136], bound flags
                                        //   by pair (int) [128], the waves' pieces of the pivot row [8][16], the decision [8]
constexpr int OFF_BIG = 2904;
constexpr int CLD = 34;                 // doubles per (i & 3) run of the column buffer (32 + 2: the four runs on different banks)
__host__ __device__ constexpr int wg2_lds_doubles(int n, int m, int pad)
{
    int big = 2 * 4 * VLD;
    if (pad * pad > big) big = pad * pad;
    if (2 * 16 * (pad + 2) > big) big = 2 * 16 * (pad + 2);
    if (n * (m | 1) > big) big = n * (m | 1);
    return OFF_BIG + big;
}

// sixteen-way scalar dispatch on a wave-uniform index (0..15; anything else: nothing): leaf k names register k statically
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

#define DISPATCH16_X(...) DISPATCH16(__VA_ARGS__)       /* (arguments expanded first: LEAVES16 below yields sixteen of them) */

// NR: row tiles = column tiles of the class (n, m <= 16 NR; sizes inside a class are padded: identity rows in H, zero rows /
// columns elsewhere); the workgroup has 2 NR wavefronts
template <int NR>
__global__ __launch_bounds__(128 * NR, (NR >= 7) ? 4 : 3) void schur_wg2_nodes(AviBatchArgs a)
{
    constexpr int pad = 16 * NR;
    // schedule hint of a resident handle (longest first): this workgroup's node; a bad entry leaves the slot idle (uniform exit)
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if ((unsigned)b >= (unsigned)a.batch) return;
    // Which hardware wave plays which part: the wavefronts of a workgroup that has the CU to itself go to the SIMDs round-robin,
    // so the hardware waves 2, 6, 10, 14 share a SIMD.  For NR <= 6 they all become C waves -- the first of them the leader --, so
    // that no H wave's exchange runs on the SIMD on which the leader takes its turn: +3 % at 65 .. 80, +2 % at 96 (with 14 or 16
    // waves the H waves would crowd on three SIMDs: -0.5 % at 112, nothing at 128 -- those keep the plain order).
    const int hwv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int v = hwv;
    if constexpr (NR <= 6) {
        constexpr int nq = (2 * NR + 1) / 4;                 // hardware waves == 2 (mod 4) among the 2 NR
        if ((hwv & 3) == 2) v = NR + (hwv >> 2);             // C waves 0 .. nq - 1
        else {
            const int rank = hwv - ((hwv + 1) >> 2);         // its position among the others
            v = rank < NR ? rank : NR + nq + (rank - NR);
        }
    }
    const bool isH = v < NR;                            // H wave (row tile v) or C wave (row tile v - NR)
    const int I = isH ? v : v - NR;
    int l = (int)threadIdx.x & 63, lc = l & 15, lq = l >> 4, tid = 64 * v + l;
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    extern __shared__ __attribute__((aligned(32))) double sm[];
    double *const sQ = sm + OFF_Q, *const sU0 = sm + OFF_U + 64 * I, *const sPr = sm + OFF_PR;
    double *const sRed = sm + OFF_RED, *const sH = sm + OFF_H, *const sV = sm + OFF_BIG, *const sW = sm + OFF_BIG;
    double *const colU = sm + OFF_COL, *const colN = sm + OFF_COL + 8 * CLD;      // (see the LDS map)
    double *const sCv = sm + OFF_U;                     // c = b - Ad h on its way to the leader (over Stage-A scratch)   [128]

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
    // ---- load ---------------------------------------------------------------------------------------------------------
    // H(r, c) = Qd[c * n + r] (padded rows: identity), C~(r, k) = Ad[r * m + k] (Ad is m x n column-major)
    const d4 z4 = {0.0, 0.0, 0.0, 0.0};
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
    // equality GAVI rows need their multiplier crashed in: left to the general kernel
    bool eqrow = false;
    if (tid < m) eqrow = a.nd.l[(size_t)b * m + tid] == a.nd.u[(size_t)b * m + tid];
    // Qd through LDS in panels of 16 columns: wave v takes columns v and v + 2 NR of a panel (lane <-> rows l, l + 64:
    // coalesced), two panels in flight; H wave I reads tile (I, J) out of panel J.  (Run by both programs below, each with its
    // own copy: READ names the tile of an H wave, or nothing.)
    constexpr int LDQ = pad + 2;
    double *const sQd = sm + OFF_BIG;
    constexpr int NWV = 2 * NR;
    const int c0 = v, c1 = v + NWV;                                    // this wave's columns of a panel (c1 < 16 only for NR < 8)
    double mabs = 0.0;
    auto issue = [&](int Jp, double (&pf)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cc = 16 * Jp + ((q & 2) ? c1 : c0), rr = l + 64 * (q & 1);
            const bool ok = ((q & 2) ? c1 < 16 : c0 < 16) && cc < n && rr < n;
            const double t_ = Q_[ok ? (size_t)cc * n + rr : 0];
            pf[q] = ok ? t_ : 0.0;
        }
    };
    auto park = [&](int Jp, const double (&pf)[4]) {
        double *const pb = sQd + (Jp & 1) * 16 * LDQ;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cp = (q & 2) ? c1 : c0, cc = 16 * Jp + cp, rr = l + 64 * (q & 1);
            mabs = max_abs_nc(mabs, pf[q]);
            if (cp < 16 && rr < pad) pb[cp * LDQ + rr] = (cc == rr && rr >= n) ? 1.0 : pf[q];           // padded rows: identity
        }
    };
#define M_PANEL(J, CUR, NXT, READ)                                                                  \
    if constexpr ((J) < NR) {                                                                       \
        if constexpr ((J) + 3 < NR) issue((J) + 3, NXT);                                            \
        park((J), CUR);                                                                             \
        __syncthreads();                                                                            \
        { const double *const pb = sQd + ((J) & 1) * 16 * LDQ; (void)pb; READ }                     \
    }
#define M_TILE_OF(T) _Pragma("unroll") for (int g = 0; g < 4; ++g) T[g] = pb[lc * LDQ + 16 * I + 4 * g + lq];
    // max |M| over the records, then the no-pivoting threshold; a node with an equality row is declined (uniform over the
    // workgroup: both programs take the same way out)
#define M_SCALE                                                                                     \
    {                                                                                               \
        double r = mabs;                                                                            \
        _Pragma("unroll") for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(r, off, 64); r = o > r ? o : r; } \
        if (l == 0) sRed[v] = r;                                                                    \
    }                                                                                               \
    const bool declined = __syncthreads_or(eqrow ? 1 : 0) != 0;                                     \
    double mscale = sRed[0];                                                                        \
    _Pragma("unroll") for (int k = 1; k < 2 * NR; ++k) mscale = fmax(mscale, sRed[k]);              \
    const double diag_thr = udbl(1e-4 * (mscale > 1.0 ? mscale : 1.0));

    // ---- rank-4 block pivots ---------------------------------------------------------------------------------
    // (the H waves and the C waves run their own straight-line copies of the steps -- the same barriers in both --, so that
    //  every tile stays in its registers: no role test, and no register shuffling around one, inside a step)
    bool fail = false, any_decline = false;
#define M_PUBT(J, T, GP) if constexpr ((J) < NR) sVp[lq * VLD + 16 * (J) + lc] = T[GP];
#define M_PUBC(J, T, GP) if constexpr ((J) < NR) sVp[lq * VLD + 128 + 16 * (J) + lc] = T[GP];
#define M_PUB_H(KB, THJP)                                                                           \
    if ((KB) / 4 < NR && !fail && 4 * (KB) < n) {                                                   \
        constexpr int JP = (KB) / 4, GP = (KB) % 4, cq = 4 * GP, par = (KB) & 1;                    \
        double *const sVp = sV + par * 4 * VLD;                                                     \
        double *const sPp = sPr + par * 24;                                                         \
        double *const sUI = sU0 + par * 512;                                                        \
        const int kcol = lc - cq;                                                                   \
        if (kcol >= 0 && kcol < 4) {                                                                \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                         \
                double val = THJP[g];                                                               \
                if (I == JP && g == GP) { sPp[lq * 4 + kcol] = val; if (lq == kcol) val -= 1.0; }   \
                sUI[(4 * g + lq) * 4 + kcol] = val;                                                 \
            }                                                                                       \
        }                                                                                           \
        if (I == JP) {          /* the owner of the pivot rows of H */                              \
            if constexpr (JP <= 0) M_PUBT(0, th0, GP)                                               \
            if constexpr (JP <= 1) M_PUBT(1, th1, GP)                                               \
            if constexpr (JP <= 2) M_PUBT(2, th2, GP)                                               \
            if constexpr (JP <= 3) M_PUBT(3, th3, GP)                                               \
            if constexpr (JP <= 4) M_PUBT(4, th4, GP)                                               \
            if constexpr (JP <= 5) M_PUBT(5, th5, GP)                                               \
            if constexpr (JP <= 6) M_PUBT(6, th6, GP)                                               \
            M_PUBT(7, th7, GP)                                                                      \
            if (l >= cq && l < cq + 4) sPp[16 + l - cq] = kx;       /* their extra-column entries x_piv */ \
        }                                                                                           \
        __syncthreads();                                                                            \
    }
#define M_PUB_C(KB, THJP)                                                                           \
    if ((KB) / 4 < NR && !fail && 4 * (KB) < n) {                                                   \
        constexpr int JP = (KB) / 4, GP = (KB) % 4, par = (KB) & 1;                                 \
        double *const sVp = sV + par * 4 * VLD;                                                     \
        if (I == JP) {          /* the owner of the pivot rows of C~ */                             \
            M_PUBC(0, tc0, GP) M_PUBC(1, tc1, GP) M_PUBC(2, tc2, GP) M_PUBC(3, tc3, GP)             \
            M_PUBC(4, tc4, GP) M_PUBC(5, tc5, GP) M_PUBC(6, tc6, GP) M_PUBC(7, tc7, GP)             \
        }                                                                                           \
        __syncthreads();                                                                            \
    }
#define M_UPD(J, T) if constexpr ((J) < NR) { const double vr_ = sVp[lq * VLD + 16 * (J) + lc]; T = MFMA_NEGA(au, vr_, T); }
#define M_UPDC(J, T) if constexpr ((J) < NR) { const double vr_ = sVp[lq * VLD + 128 + 16 * (J) + lc]; T = MFMA_NEGA(au, vr_, T); }
    // P = L U (unit lower L, no pivoting; uniform: every lane of every wave), then column lq of P^-1 (L y = e_lq, U x = y):
    // U' = U P^-1 straight in the A-operand layout
#define M_FACTOR(KB)                                                                                \
        constexpr int JP = (KB) / 4, par = (KB) & 1;                                                \
        const double *const sVp = sV + par * 4 * VLD;                                               \
        const double *const sPp = sPr + par * 24;                                                   \
        const double *const sUI = sU0 + par * 512;                                                  \
        double pm[4][4];                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
            const d4 row = *reinterpret_cast<const d4 *>(sPp + i * 4);                              \
            pm[i][0] = row[0]; pm[i][1] = row[1]; pm[i][2] = row[2]; pm[i][3] = row[3];             \
        }                                                                                           \
        bool okp = true;                                                                            \
        double rd[4];                                                                               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
            okp = okp & (fabs(pm[s][s]) >= diag_thr);                                               \
            rd[s] = rcp64(pm[s][s]);                                                                \
            _Pragma("unroll") for (int i = s + 1; i < 4; ++i) {                                     \
                const double f = pm[i][s] * rd[s];                                                  \
                pm[i][s] = f;                                                                       \
                _Pragma("unroll") for (int j = s + 1; j < 4; ++j) pm[i][j] = fma(-f, pm[s][j], pm[i][j]); \
            }                                                                                       \
        }                                                                                           \
        if (!ubool(okp)) { fail = true; }
#define M_UPRIME                                                                                    \
            const double e0 = lq == 0 ? 1.0 : 0.0, e1 = lq == 1 ? 1.0 : 0.0, e2 = lq == 2 ? 1.0 : 0.0, e3 = lq == 3 ? 1.0 : 0.0; \
            const double y1 = fma(-pm[1][0], e0, e1);                                               \
            const double y2 = fma(-pm[2][1], y1, fma(-pm[2][0], e0, e2));                           \
            const double y3 = fma(-pm[3][2], y2, fma(-pm[3][1], y1, fma(-pm[3][0], e0, e3)));       \
            const double p3 = y3 * rd[3];                                                           \
            const double p2 = fma(-pm[2][3], p3, y2) * rd[2];                                       \
            const double p1 = fma(-pm[1][3], p3, fma(-pm[1][2], p2, y1)) * rd[1];                   \
            const double p0 = fma(-pm[0][3], p3, fma(-pm[0][2], p2, fma(-pm[0][1], p1, e0))) * rd[0]; \
            const d4 ur = *reinterpret_cast<const d4 *>(sUI + lc * 4);                              \
            const double au = fma(ur[3], p3, fma(ur[2], p2, fma(ur[1], p1, ur[0] * p0)));
#define M_ELIM_H(KB)                                                                                \
    if ((KB) / 4 < NR && !fail && 4 * (KB) < n) {                                                   \
        M_FACTOR(KB)                                                                                \
        else {                                                                                      \
            M_UPRIME                                                                                \
            kx -= xsum_rows(au * sPp[16 + lq]);                                                     \
            if constexpr (JP <= 0) M_UPD(0, th0)                                                    \
            if constexpr (JP <= 1) M_UPD(1, th1)                                                    \
            if constexpr (JP <= 2) M_UPD(2, th2)                                                    \
            if constexpr (JP <= 3) M_UPD(3, th3)                                                    \
            if constexpr (JP <= 4) M_UPD(4, th4)                                                    \
            if constexpr (JP <= 5) M_UPD(5, th5)                                                    \
            if constexpr (JP <= 6) M_UPD(6, th6)                                                    \
            M_UPD(7, th7)                                                                           \
            wave_sync();                                                                            \
        }                                                                                           \
    }
#define M_ELIM_C(KB)                                                                                \
    if ((KB) / 4 < NR && !fail && 4 * (KB) < n) {                                                   \
        M_FACTOR(KB)                                                                                \
        else {                                                                                      \
            (void)JP;                                                                               \
            M_UPRIME                                                                                \
            M_UPDC(0, tc0) M_UPDC(1, tc1) M_UPDC(2, tc2) M_UPDC(3, tc3)                             \
            M_UPDC(4, tc4) M_UPDC(5, tc5) M_UPDC(6, tc6) M_UPDC(7, tc7)                             \
            wave_sync();                                                                            \
        }                                                                                           \
    }
#define M_STEP4(R, K0, TH) M_PUB_##R(K0, TH) M_ELIM_##R(K0) M_PUB_##R(K0 + 1, TH) M_ELIM_##R(K0 + 1)   \
                           M_PUB_##R(K0 + 2, TH) M_ELIM_##R(K0 + 2) M_PUB_##R(K0 + 3, TH) M_ELIM_##R(K0 + 3)
    // Ad for the post-check: requested (coalesced) and parked in LDS by all waves, each role from its own program
#define M_STAGE_AD                                                                                  \
    {                                                                                               \
        constexpr int TPB_ = 128 * NR;                                                              \
        for (int e0 = tid; e0 < m * n; e0 += 4 * TPB_) {                                            \
            double vv[4];                                                                           \
            _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) { const int e = e0 + q4 * TPB_; vv[q4] = e < m * n ? A_[e] : 0.0; } \
            _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                      \
                const int e = e0 + q4 * TPB_;                                                       \
                const int j = e / m, i = e - j * m;                                                 \
                if (e < m * n) sAd[j * LDA + i] = vv[q4];                                           \
            }                                                                                       \
        }                                                                                           \
    }
    // ---- from here on the H waves and the C waves run TWO PROGRAMS, barrier for barrier the same: what belongs to one role
    // (H tiles, S, the dictionary, the Lemke state / the W~ tiles) is never live in the other role's code
    constexpr int NBP = 128, XC = 128, VTH = 256;
    constexpr int CODE_PIVOT = 0, CODE_FLIP = 1, CODE_STOP = 2;
    const int LDA = m | 1;
    double *const sAd = sm + OFF_BIG, *const sX = sm + OFF_BIG;
    double *const sDecD = sm + OFF_B + 904;                            // the leader's decision: 1 / pivot                [1]
    int *const sDecI = reinterpret_cast<int *>(sm + OFF_B + 906);      // ... what, row, next column                      [4]
    double *const sval = sm + OFF_U, *const sz = sm + OFF_U + 264;     // read-back: values by variable id [264], z in item order [256]
    typedef const AviBatchArgs __attribute__((address_space(4))) *kargs_t;
    int status = QPN_FAILURE;
    int pivots = n;                       // the crash brings n free variables in (Stage A)
    if (isH) {
        // =============================================== the H-wave program ===============================================
        d4 th0 = z4, th1 = z4, th2 = z4, th3 = z4, th4 = z4, th5 = z4, th6 = z4, th7 = z4;      // row tile I of H
        {
            // (four register sets: panels J + 1 .. J + 3 are in flight while panel J is parked and read)
            double pa[4], pb_[4], pc[4], pd[4];
            issue(0, pa); issue(1, pb_); issue(2, pc);
            M_PANEL(0, pa, pd, M_TILE_OF(th0)) M_PANEL(1, pb_, pa, M_TILE_OF(th1)) M_PANEL(2, pc, pb_, M_TILE_OF(th2))
            M_PANEL(3, pd, pc, M_TILE_OF(th3)) M_PANEL(4, pa, pd, M_TILE_OF(th4)) M_PANEL(5, pb_, pa, M_TILE_OF(th5))
            M_PANEL(6, pc, pb_, M_TILE_OF(th6)) M_PANEL(7, pd, pc, M_TILE_OF(th7))
        }
        M_SCALE
        any_decline = declined;
        double kx = (l < 16 && 16 * I + l < n) ? sQ[16 * I + l] : 0.0;       // lane l <-> row 16 I + l of the extra column
        __syncthreads();                    // the staged Qd has been read: the published pivot rows reuse its area
        if (declined) fail = true;          // (the steps below, and everything behind them, are skipped)
        STAMP(0);   // load
        M_STEP4(H, 0, th0) M_STEP4(H, 4, th1) M_STEP4(H, 8, th2) M_STEP4(H, 12, th3)
        M_STEP4(H, 16, th4) M_STEP4(H, 20, th5) M_STEP4(H, 24, th6) M_STEP4(H, 28, th7)
        STAMP(1);   // stage A
        if (!fail) {
        l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
        __syncthreads();                               // X1: the last step's pivot rows have been read (sW reuses them)
        const int rot = lq & 1;
        if (l < 16) sH[16 * I + l] = kx;
        __syncthreads();                               // X2: W~ and h are in LDS
        // ---- S = Ad W~ (row tile I), c = b - Ad h
        d4 s0 = z4, s1 = z4, s2 = z4, s3 = z4, s4 = z4, s5 = z4, s6 = z4, s7 = z4, sx = z4;
        {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ri = 16 * I + 4 * g + lq;
                sx[g] = (lc == 0 && ri < m) ? sQ[n + ri] : 0.0;
            }
            const int arow = 16 * I + lc;
            auto jr = [&](int Jc) { int j = Jc + rot; return j >= NR ? j - NR : j; };
            const int j0 = jr(0), j1 = jr(1), j2 = jr(2), j3 = jr(3), j4 = jr(4), j5 = jr(5), j6 = jr(6), j7 = jr(7);
            // A operands (16 x 4 blocks of Ad, element (i = lc, k = lq)) in batches of 8 k-blocks, the next batch requested
            // before the MFMAs of the current one: the round trips to L2 hide behind 8 x (NR + 1) MFMAs
            auto aload = [&](int kk) -> double {
                const int r = 4 * kk + lq;                              // k index: column of Ad, row of W~
                const bool valid = kk < 4 * NR && r < n && arow < m;
                const double t_ = A_[valid ? (size_t)r * m + arow : 0];
                return valid ? t_ : 0.0;
            };
            double ab[8], an[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ab[q] = aload(q);
#pragma unroll
            for (int k0 = 0; k0 < 4 * NR; k0 += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) an[q] = aload(k0 + 8 + q);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int kk = k0 + q;
                    if (kk < 4 * NR) {
                        const int r = 4 * kk + lq;
                        const double a_ = ab[q];
                        const double *wr = sW + r * pad + lc;
                        s0 = MFMA(a_, wr[16 * j0], s0);
                        s1 = MFMA(a_, wr[16 * j1], s1);
                        s2 = MFMA(a_, wr[16 * j2], s2);
                        s3 = MFMA(a_, wr[16 * j3], s3);
                        s4 = MFMA(a_, wr[16 * j4], s4);
                        if constexpr (NR > 5) s5 = MFMA(a_, wr[16 * j5], s5);
                        if constexpr (NR > 6) s6 = MFMA(a_, wr[16 * j6], s6);
                        if constexpr (NR > 7) s7 = MFMA(a_, wr[16 * j7], s7);
                        const double hb = lc == 0 ? sH[r] : 0.0;
                        sx = MFMA_NEGA(a_, hb, sx);
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) ab[q] = an[q];
            }
        }
        __syncthreads();                               // X3: W~ in LDS has been read: S goes through the same area
        // ---- S from row tiles to COLUMN tiles: H wave I hands tile (I, J) to H wave J; c goes to the leader
        {
#define M_SOUT(J, T)                                                                                \
    if constexpr ((J) < NR) {                                                                       \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) sX[((I * NR + (J)) * 4 + g) * 64 + l] = T[g]; \
    }
            M_SOUT(0, s0) M_SOUT(1, s1) M_SOUT(2, s2) M_SOUT(3, s3) M_SOUT(4, s4) M_SOUT(5, s5) M_SOUT(6, s6) M_SOUT(7, s7)
#undef M_SOUT
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (lc == 0) sCv[16 * I + 4 * g + lq] = sx[g];
        }
        __syncthreads();                               // X4
        // the dictionary tiles of H wave I -- column tile I, rows 16 R + 4 g + lq, column 16 I + lc at register 4 R + g --: NAMED
        // scalars (never an array: every dynamic select of one of them happens inside an asm dispatch on a scalar index)
#define TD(j) td_##j
#define FOR_T(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)                             \
                 M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#define M_TLOAD(j) double TD(j) = ((j) < 4 * NR) ? sX[((((j) >> 2) * NR + I) * 4 + ((j) & 3)) * 64 + l] : 0.0;
        FOR_T(M_TLOAD)
#undef M_TLOAD
        __syncthreads();                               // X5: S has been read: Ad for the post-check goes into the same area
        l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
        M_STAGE_AD
        // ================= Stage B: Lemke on the Schur dictionary =================
        // pair k (k < m) <-> item row n + k.  ids: p_k -> k, d_k -> 128 + k, artificial -> 256; column index 128 = the extra
        // (covering) column.  The LEADER is C wave 0 (wave NR): it keeps the bookkeeping and runs the ratio test; the H waves
        // hold the dictionary.  Per pivot t:
        //   barrier B(t): the leader's decision {what, row r, 1 / pivot, column c, next column} is posted; the owner of the NEXT
        //   column publishes it AS IT STANDS (before the exchange) -- barrier A'(t) -- the H waves run the exchange of pivot t
        //   while the leader derives the next column from the old one (T'[i][j] = T[i][j] - u_i pv_j, row r: -pv_j: the
        //   same two operations on the same bits as the tiles' own update) and takes its next turn -- barrier B(t+1).
        // So the exchange (4 NR v_fma_f64 + fix-ups + two register dispatches) and the leader's turn overlap.
        double *const rowbuf = sm + OFF_B + 776 + 16 * I;                  // this H wave's piece of the pivot row            [16]
        __syncthreads();                               // X6 = B(0): the leader's first decision is posted
        STAMP(4);   // W~ hand-over, S product, tile hand-over, Ad staging
        int code = uni(sDecI[0]);
#define T_OPS_RW_A [t0] "+v"(TD(0)), [t1] "+v"(TD(1)), [t2] "+v"(TD(2)), [t3] "+v"(TD(3)), [t4] "+v"(TD(4)), [t5] "+v"(TD(5)),     \
                       [t6] "+v"(TD(6)), [t7] "+v"(TD(7)), [t8] "+v"(TD(8)), [t9] "+v"(TD(9)), [t10] "+v"(TD(10)), [t11] "+v"(TD(11)), \
                       [t12] "+v"(TD(12)), [t13] "+v"(TD(13)), [t14] "+v"(TD(14)), [t15] "+v"(TD(15))
#define T_OPS_RW_B [t0] "+v"(TD(16)), [t1] "+v"(TD(17)), [t2] "+v"(TD(18)), [t3] "+v"(TD(19)), [t4] "+v"(TD(20)), [t5] "+v"(TD(21)), \
                       [t6] "+v"(TD(22)), [t7] "+v"(TD(23)), [t8] "+v"(TD(24)), [t9] "+v"(TD(25)), [t10] "+v"(TD(26)), [t11] "+v"(TD(27)), \
                       [t12] "+v"(TD(28)), [t13] "+v"(TD(29)), [t14] "+v"(TD(30)), [t15] "+v"(TD(31))
#define T_OPS_R_A [t0] "v"(TD(0)), [t1] "v"(TD(1)), [t2] "v"(TD(2)), [t3] "v"(TD(3)), [t4] "v"(TD(4)), [t5] "v"(TD(5)),           \
                      [t6] "v"(TD(6)), [t7] "v"(TD(7)), [t8] "v"(TD(8)), [t9] "v"(TD(9)), [t10] "v"(TD(10)), [t11] "v"(TD(11)),       \
                      [t12] "v"(TD(12)), [t13] "v"(TD(13)), [t14] "v"(TD(14)), [t15] "v"(TD(15))
#define T_OPS_R_B [t0] "v"(TD(16)), [t1] "v"(TD(17)), [t2] "v"(TD(18)), [t3] "v"(TD(19)), [t4] "v"(TD(20)), [t5] "v"(TD(21)),     \
                      [t6] "v"(TD(22)), [t7] "v"(TD(23)), [t8] "v"(TD(24)), [t9] "v"(TD(25)), [t10] "v"(TD(26)), [t11] "v"(TD(27)),   \
                      [t12] "v"(TD(28)), [t13] "v"(TD(29)), [t14] "v"(TD(30)), [t15] "v"(TD(31))
#define LEAVES16(OP) OP "%[t0]", OP "%[t1]", OP "%[t2]", OP "%[t3]", OP "%[t4]", OP "%[t5]", OP "%[t6]", OP "%[t7]",                \
                         OP "%[t8]", OP "%[t9]", OP "%[t10]", OP "%[t11]", OP "%[t12]", OP "%[t13]", OP "%[t14]", OP "%[t15]"
        while (code != CODE_STOP) {
            const int r = uni(sDecI[1]), c = uni(sDecI[2]), cnext = uni(sDecI[3]), par = uni(sDecI[4]);
            const double inv = udbl(sDecD[0]);
            // ---- the NEXT entering column as it stands, published by the 4 lanes that hold it: row i goes to [i & 3][i >> 2]
            if (cnext != XC && I == (cnext >> 4) && lc == (cnext & 15)) {
                d4 *const dst = reinterpret_cast<d4 *>(colN + lq * CLD);
                dst[0] = d4{TD(0), TD(1), TD(2), TD(3)};
                dst[1] = d4{TD(4), TD(5), TD(6), TD(7)};
                dst[2] = d4{TD(8), TD(9), TD(10), TD(11)};
                dst[3] = d4{TD(12), TD(13), TD(14), TD(15)};
                dst[4] = d4{TD(16), TD(17), TD(18), TD(19)};
                if constexpr (NR > 5) dst[5] = d4{TD(20), TD(21), TD(22), TD(23)};
                if constexpr (NR > 6) dst[6] = d4{TD(24), TD(25), TD(26), TD(27)};
                if constexpr (NR > 7) dst[7] = d4{TD(28), TD(29), TD(30), TD(31)};
            }
            __syncthreads();                                // barrier A'
            STAMP(5);   // decision read + next column + barrier A'
            // ---- the exchange (a flip runs it with empty lane masks and a zero row: a no-op, so the tile registers have one
            // definition per iteration)
            {
                const bool piv = code == CODE_PIVOT;
                const int rq = r & 3, rsel = piv ? (r >> 2) : 32;
                const bool own = piv && c != XC && I == (c >> 4);
                const unsigned long long mrow = piv ? 0xFFFFull << (16 * rq) : 0ull;
                const unsigned long long mcol = own ? 0x0001000100010001ull << (c & 15) : 0ull;
                const int csA = uni(rsel < 16 ? rsel : 16), csB = uni((rsel >= 16 && rsel < 32) ? rsel - 16 : 16);
                // this wave's piece of the pivot row: the 16 lanes that hold it pick the register under the scalar dispatch
                {
                    const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(rowbuf + lc);
                    asm volatile("s_mov_b64 exec, %[mr]\n\t"
                                 DISPATCH16_X("Lw2ra", LEAVES16("ds_write_b64 %[ad], "))
                                 "s_mov_b64 exec, -1"
                                 : : T_OPS_R_A, [cs] "s"(csA), [mr] "s"(mrow), [ad] "v"(ad) : "scc", "memory");
                    asm volatile("s_mov_b64 exec, %[mr]\n\t"
                                 DISPATCH16_X("Lw2rb", LEAVES16("ds_write_b64 %[ad], "))
                                 "s_mov_b64 exec, -1"
                                 : : T_OPS_R_B, [cs] "s"(csB), [mr] "s"(mrow), [ad] "v"(ad) : "scc", "memory");
                    // the entry of the pivot column itself is replaced by -1, so that row * inv carries -inv there
                    if (own && l == 16 * rq + (c & 15)) rowbuf[c & 15] = -1.0;
                }
                wave_sync();
                const double pv = piv ? rowbuf[lc] * inv : 0.0;
                // this lane's column entries: rows lq + 4 k, k = 4 R + g
                const d4 *const up = reinterpret_cast<const d4 *>(colU + par * 4 * CLD + lq * CLD);
                const double inv_s = udbl(inv);
#define M_XCHG4(A0, A1, A2, A3, U)                                                                                                 \
                asm volatile("v_fma_f64 %[a0], -%[u0], %[pv], %[a0]\n\tv_fma_f64 %[a1], -%[u1], %[pv], %[a1]\n\t"                      \
                             "v_fma_f64 %[a2], -%[u2], %[pv], %[a2]\n\tv_fma_f64 %[a3], -%[u3], %[pv], %[a3]\n\t"                      \
                             /* column c: T[i][c] = u_i * inv on the 4 lanes that own it */                                            \
                             "s_mov_b64 exec, %[mc]\n\ts_cbranch_execz 31f\n\t"                                                  \
                             "v_mul_f64 %[a0], %[u0], %[iv]\n\tv_mul_f64 %[a1], %[u1], %[iv]\n\t"                                      \
                             "v_mul_f64 %[a2], %[u2], %[iv]\n\tv_mul_f64 %[a3], %[u3], %[iv]\n"                                        \
                             "31:\n\ts_mov_b64 exec, -1"                                                                        \
                             : [a0] "+v"(TD(A0)), [a1] "+v"(TD(A1)), [a2] "+v"(TD(A2)), [a3] "+v"(TD(A3))                              \
                             : [u0] "v"(U[0]), [u1] "v"(U[1]), [u2] "v"(U[2]), [u3] "v"(U[3]), [pv] "v"(pv), [iv] "s"(inv_s), [mc] "s"(mcol));
                { const d4 u0 = up[0], u1 = up[1], u2 = up[2], u3 = up[3];
                  M_XCHG4(0, 1, 2, 3, u0) M_XCHG4(4, 5, 6, 7, u1) M_XCHG4(8, 9, 10, 11, u2) M_XCHG4(12, 13, 14, 15, u3) }
                { const d4 u4 = up[4];
                  M_XCHG4(16, 17, 18, 19, u4)
                  if constexpr (NR > 5) { const d4 u5 = up[5]; M_XCHG4(20, 21, 22, 23, u5) }
                  if constexpr (NR > 6) { const d4 u6 = up[6]; M_XCHG4(24, 25, 26, 27, u6) }
                  if constexpr (NR > 7) { const d4 u7 = up[7]; M_XCHG4(28, 29, 30, 31, u7) } }
#undef M_XCHG4
                // row r: T[r][j] = -pv_j on the 16 lanes that own it (pv carries -inv at column c)
                asm volatile("s_mov_b64 exec, %[mr]\n\t"
                             DISPATCH16("Lw2fa", "v_mul_f64 %[t0], %[pv], -1.0", "v_mul_f64 %[t1], %[pv], -1.0", "v_mul_f64 %[t2], %[pv], -1.0",
                                        "v_mul_f64 %[t3], %[pv], -1.0", "v_mul_f64 %[t4], %[pv], -1.0", "v_mul_f64 %[t5], %[pv], -1.0",
                                        "v_mul_f64 %[t6], %[pv], -1.0", "v_mul_f64 %[t7], %[pv], -1.0", "v_mul_f64 %[t8], %[pv], -1.0",
                                        "v_mul_f64 %[t9], %[pv], -1.0", "v_mul_f64 %[t10], %[pv], -1.0", "v_mul_f64 %[t11], %[pv], -1.0",
                                        "v_mul_f64 %[t12], %[pv], -1.0", "v_mul_f64 %[t13], %[pv], -1.0", "v_mul_f64 %[t14], %[pv], -1.0",
                                        "v_mul_f64 %[t15], %[pv], -1.0")
                             "s_mov_b64 exec, -1"
                             : T_OPS_RW_A : [cs] "s"(csA), [mr] "s"(mrow), [pv] "v"(pv) : "scc");
                asm volatile("s_mov_b64 exec, %[mr]\n\t"
                             DISPATCH16("Lw2fb", "v_mul_f64 %[t0], %[pv], -1.0", "v_mul_f64 %[t1], %[pv], -1.0", "v_mul_f64 %[t2], %[pv], -1.0",
                                        "v_mul_f64 %[t3], %[pv], -1.0", "v_mul_f64 %[t4], %[pv], -1.0", "v_mul_f64 %[t5], %[pv], -1.0",
                                        "v_mul_f64 %[t6], %[pv], -1.0", "v_mul_f64 %[t7], %[pv], -1.0", "v_mul_f64 %[t8], %[pv], -1.0",
                                        "v_mul_f64 %[t9], %[pv], -1.0", "v_mul_f64 %[t10], %[pv], -1.0", "v_mul_f64 %[t11], %[pv], -1.0",
                                        "v_mul_f64 %[t12], %[pv], -1.0", "v_mul_f64 %[t13], %[pv], -1.0", "v_mul_f64 %[t14], %[pv], -1.0",
                                        "v_mul_f64 %[t15], %[pv], -1.0")
                             "s_mov_b64 exec, -1"
                             : T_OPS_RW_B : [cs] "s"(csB), [mr] "s"(mrow), [pv] "v"(pv) : "scc");
            }
            STAMP(7);   // the exchange
            __syncthreads();                                // barrier B
            STAMP(6);   // waiting for the leader's decision
            code = uni(sDecI[0]);
        }
        __syncthreads();                                // every wave is out of the loop
        __syncthreads();                                // the leader has posted the values by variable id
        }
#undef LEAVES16
#undef T_OPS_R_B
#undef T_OPS_R_A
#undef T_OPS_RW_B
#undef T_OPS_RW_A
#undef FOR_T
#undef TD
    } else {
        // =============================================== the C-wave program ===============================================
        // C~ tiles: direct loads (a row group of a tile is 128 contiguous bytes of Ad)
        d4 tc0 = z4, tc1 = z4, tc2 = z4, tc3 = z4, tc4 = z4, tc5 = z4, tc6 = z4, tc7 = z4;      // row tile I of C~, later W~
#define M_LOADC(J, T)                                                                               \
    if constexpr ((J) < NR) {                                                                       \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
            const int rr = 16 * I + 4 * g + lq, ck = 16 * (J) + lc;                                 \
            const bool valid = rr < n && ck < m;                                                    \
            const double t_ = A_[valid ? (size_t)rr * m + ck : 0];                                  \
            T[g] = valid ? t_ : 0.0;                                                                \
        }                                                                                           \
    }
        M_LOADC(0, tc0) M_LOADC(1, tc1) M_LOADC(2, tc2) M_LOADC(3, tc3) M_LOADC(4, tc4) M_LOADC(5, tc5) M_LOADC(6, tc6) M_LOADC(7, tc7)
#undef M_LOADC
        {
            double pa[4], pb_[4], pc[4], pd[4];
            issue(0, pa); issue(1, pb_); issue(2, pc);
            M_PANEL(0, pa, pd, ) M_PANEL(1, pb_, pa, ) M_PANEL(2, pc, pb_, ) M_PANEL(3, pd, pc, )
            M_PANEL(4, pa, pd, ) M_PANEL(5, pb_, pa, ) M_PANEL(6, pc, pb_, ) M_PANEL(7, pd, pc, )
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mabs = max_abs_nc(max_abs_nc(max_abs_nc(max_abs_nc(mabs, tc0[g]), tc1[g]), tc2[g]), tc3[g]);
            mabs = max_abs_nc(max_abs_nc(max_abs_nc(max_abs_nc(mabs, tc4[g]), tc5[g]), tc6[g]), tc7[g]);
        }
        M_SCALE
        any_decline = declined;
        __syncthreads();                    // the staged Qd has been read: the published pivot rows reuse its area
        if (declined) fail = true;
        M_STEP4(C, 0, tc0) M_STEP4(C, 4, tc1) M_STEP4(C, 8, tc2) M_STEP4(C, 12, tc3)
        M_STEP4(C, 16, tc4) M_STEP4(C, 20, tc5) M_STEP4(C, 24, tc6) M_STEP4(C, 28, tc7)
        if (!fail) {
        l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
        __syncthreads();                               // X1
        {
            const int rot = lq & 1;                    // rows of odd lq store their column blocks rotated by one: the two row
                                                       // groups of a half-wave read disjoint banks
#define M_WOUT(J, T)                                                                                \
    if constexpr ((J) < NR) {                                                                       \
        int jj = (J) + rot; if (jj >= NR) jj -= NR;                                                 \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) sW[(16 * I + 4 * g + lq) * pad + 16 * jj + lc] = T[g]; \
    }
            M_WOUT(0, tc0) M_WOUT(1, tc1) M_WOUT(2, tc2) M_WOUT(3, tc3) M_WOUT(4, tc4) M_WOUT(5, tc5) M_WOUT(6, tc6) M_WOUT(7, tc7)
#undef M_WOUT
        }
        __syncthreads();                               // X2
        __syncthreads();                               // X3 (the H waves form S)
        __syncthreads();                               // X4 (... hand its tiles over)
        __syncthreads();                               // X5 (... and take them)
        l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
        M_STAGE_AD
        STAMP(4);   // (C waves) stage A is in slot 0 for them; W~ hand-over and the waits for the S product
        int code;
        if (v == NR) {
            // ================================ the LEADER (C wave 0): ratio test, bookkeeping, decisions ================================
            const bool lead = true;
            const bool act0 = l < m, act1 = 64 + l < m;
            double xb0 = act0 ? sCv[l] : 0.0, xb1 = act1 ? sCv[64 + l] : 0.0;
            double *const sLo = sm + OFF_B, *const sHi = sm + OFF_B + 128;     // fixed pair bounds (looked up by a uniform index)
            double *const sRowLo = sm + OFF_B + 256, *const sRowHi = sm + OFF_B + 384;     // interval of each row's basic variable
            double *const sNb = sm + OFF_B + 512;                              // value of each column's nonbasic variable (128: extra)
            int *const sSat = reinterpret_cast<int *>(sm + OFF_B + 648);       // pair k: 1 = bounded variable rests at its upper bound
            // (the leader keeps in REGISTERS only what every lane updates every pivot -- basic values, the extra column -- and the two
            //  id vectors it searches; bounds, nonbasic values and flags, touched at one index per pivot, live in the tables above:
            //  the wave also carries 4 NR dictionary entries per lane)
            if (lead) {
                double lo_0 = -QINF, hi_0 = QINF, lo_1 = -QINF, hi_1 = QINF;
                if (act0) { lo_0 = a.nd.l[(size_t)b * m + l]; hi_0 = a.nd.u[(size_t)b * m + l]; }
                if (act1) { lo_1 = a.nd.l[(size_t)b * m + 64 + l]; hi_1 = a.nd.u[(size_t)b * m + 64 + l]; }
                sLo[l] = lo_0; sHi[l] = hi_0; sLo[64 + l] = lo_1; sHi[64 + l] = hi_1;
                sRowLo[l] = lo_0; sRowHi[l] = hi_0; sRowLo[64 + l] = lo_1; sRowHi[64 + l] = hi_1;
                sNb[l] = 0.0; sNb[64 + l] = 0.0; if (l < 8) sNb[128 + l] = 0.0;
                sSat[l] = 0; sSat[64 + l] = 0;
            }
            int rowvar0 = act0 ? l : -1, rowvar1 = act1 ? 64 + l : -1;
            int colvar0 = act0 ? NBP + l : -1, colvar1 = act1 ? NBP + 64 + l : -1;
            int cvx = VTH;
            double tcol0 = 0.0, tcol1 = 0.0;
            const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;
            int c = XC, par = 0;
            bool sneg = true;
            double self_lim = 0.0, elo = 0.0, ehi = QINF;
            const double slack = 1e-10, ptol = a.piv_tol;
            auto col_of = [&](int var) -> int {
                int cc = wave_first(colvar0 == var);
                if (cc >= 0) return cc;
                cc = wave_first(colvar1 == var);
                return cc >= 0 ? 64 + cc : (cvx == var ? XC : -1);
            };
            if (lead) {
                const double lo_0 = sRowLo[l], hi_0 = sRowHi[l], lo_1 = sRowLo[64 + l], hi_1 = sRowHi[64 + l];
                double viol = 0.0, viol1 = 0.0;
                if (act0) viol = xb0 < lo_0 ? lo_0 - xb0 : (xb0 > hi_0 ? xb0 - hi_0 : 0.0);
                if (act1) viol1 = xb1 < lo_1 ? lo_1 - xb1 : (xb1 > hi_1 ? xb1 - hi_1 : 0.0);
                const double theta0 = wave_max_f64(fmax(viol, viol1));
                if (ubool(theta0 <= a.feas_tol)) status = QPN_SUCCESS;
                else {
                    auto cover = [&](bool act, double &xb, double lo, double hi) -> double {
                        double cov = 0.0;
                        if (!act) return 0.0;
                        if (xb < lo) {
                            double tgt = lo + (theta0 - (lo - xb));
                            if (hi < QINF) { double mid = 0.5 * (lo + hi); if (tgt > mid) tgt = mid; }
                            cov = (tgt - xb) / theta0; xb = tgt;
                        } else if (xb > hi) {
                            double tgt = hi - (theta0 - (xb - hi));
                            if (lo > -QINF) { double mid = 0.5 * (lo + hi); if (tgt < mid) tgt = mid; }
                            cov = (tgt - xb) / theta0; xb = tgt;
                        }
                        return cov;
                    };
                    tcol0 = cover(act0, xb0, lo_0, hi_0);
                    tcol1 = cover(act1, xb1, lo_1, hi_1);
                    if (l == 0) sNb[XC] = theta0;
                    self_lim = theta0;
                    status = QPN_MAX_ITERS;
                }
            }
            // the first entering column is the extra one: the leader's own
            double cm0 = tcol0, cm1 = tcol1;
            code = status == QPN_MAX_ITERS ? CODE_PIVOT : CODE_STOP;
            for (;;) {
                int dcode = code, r = 0, cnext = XC;
                double inv = 0.0;
                // what the second half of the turn needs from the first
                bool flip = false, pivot = false;
                int ve = 0, vl = 0, kf = 0;
                double rcr = 0.0, step = 0.0, leave_val = 0.0;
                if (code != CODE_STOP) {
                    // this pivot's column, for the exchange of the H waves: row i goes to [i & 3][i >> 2]
                    {
                        double *const cu = colU + par * 4 * CLD + (l & 3) * CLD + (l >> 2);
                        cu[0] = cm0; cu[16] = cm1;
                    }
                    STAMP(5);   // (leader) barriers + the next column derived
                    // ================= the leader's turn, first half: WHAT the H waves wait for =================
                    // ratio test (two-pass Harris with 1e-10 slack; largest pivot among ties, the artificial first), two rows per lane
                    const int sbit = sneg ? (int)0x80000000 : 0;
                    const double g0 = __hiloint2double(__double2hiint(cm0) ^ sbit, __double2loint(cm0));
                    const double g1 = __hiloint2double(__double2hiint(cm1) ^ sbit, __double2loint(cm1));
                    const double rc0 = rcp64(g0), rc1 = rcp64(g1);
                    // the bound the row's basic variable moves towards: from the rows' interval tables (one read per row)
                    const double tb0 = (__double2hiint(g0) < 0 ? sRowLo : sRowHi)[l], tb1 = (__double2hiint(g1) < 0 ? sRowLo : sRowHi)[64 + l];
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wbitwise-instead-of-logical"
                    const bool cnd0 = (act0 & (fabs(g0) > ptol)) & (fabs(tb0) < QINF);
                    const bool cnd1 = (act1 & (fabs(g1) > ptol)) & (fabs(tb1) < QINF);
#pragma clang diagnostic pop
                    const double dd0 = cnd0 ? (tb0 - xb0) * rc0 : QINF, dd1 = cnd1 ? (tb1 - xb1) * rc1 : QINF;
                    const double d10 = fma(slack, fabs(rc0), dd0), d11 = fma(slack, fabs(rc1), dd1);
                    const double dmax = wave_min64_with_limit_f64(min_f64_nc(d10, d11), self_lim);
                    dcode = CODE_PIVOT;
                    if (pivots >= max_piv) dcode = CODE_STOP;          // status stays MAX_ITERS
                    else if (uni(__double2hiint(dmax)) == 0x7ff00000) { status = QPN_RAY_TERM; dcode = CODE_STOP; }
                    else {
                        const unsigned long long bal0 = qpn_ballot(dd0 <= dmax), bal1 = qpn_ballot(dd1 <= dmax);
                        const int ch = c >> 6, cl = c & 63;
                        ve = (c == XC) ? cvx : (ch ? readlane_i32(colvar1, cl) : readlane_i32(colvar0, cl));     // the entering variable
                        if ((bal0 | bal1) == 0ull) {
                            // the entering variable reaches its own opposite bound first: no basis change
                            flip = true;
                            dcode = CODE_FLIP;
                            if (ve == VTH) { status = QPN_SUCCESS; dcode = CODE_STOP; }
                            else {
                                kf = ve;
                                pivots++;
                                cnext = col_of(NBP + kf);               // (the id vectors do not change in a flip)
                                if (cnext < 0) { status = QPN_FAILURE; dcode = CODE_STOP; }
                            }
                        } else {
                            pivot = true;
                            if (__popcll(bal0) + __popcll(bal1) == 1) r = bal0 ? __ffsll((long long)bal0) - 1 : 64 + __ffsll((long long)bal1) - 1;
                            else {
                                const bool ca0 = dd0 <= dmax, ca1 = dd1 <= dmax;
                                double ag0 = ca0 ? fabs(g0) : -1.0, ag1 = ca1 ? fabs(g1) : -1.0;
                                if (ca0 && rowvar0 == VTH) ag0 = QINF;
                                if (ca1 && rowvar1 == VTH) ag1 = QINF;
                                const double bestg = wave_max_f64(fmax(ag0, ag1));
                                const int r0 = wave_first(ca0 && ag0 == bestg);
                                r = r0 >= 0 ? r0 : 64 + wave_first(ca1 && ag1 == bestg);
                            }
                            r = uni(r);
                            const int rh = r >> 6, rl = r & 63;
                            rcr = rh ? readlane_f64(rc1, rl) : readlane_f64(rc0, rl);
                            inv = sneg ? -rcr : rcr;                    // 1 / T[r][c]
                            vl = rh ? readlane_i32(rowvar1, rl) : readlane_i32(rowvar0, rl);
                            step = rh ? readlane_f64(dd1, rl) : readlane_f64(dd0, rl);
                            leave_val = rh ? readlane_f64(tb1, rl) : readlane_f64(tb0, rl);
                            pivots++;
                            if (vl == VTH) { status = QPN_SUCCESS; dcode = CODE_STOP; }
                            else {
                                // the complement of the leaving variable enters next.  (colvar / cvx still hold the entering id at
                                // column c -- the write-back is in the second half --, which is never vn's; the id that lands
                                // there, vl, is never its own complement either)
                                const int vn = vl < NBP ? NBP + vl : vl - NBP;
                                cnext = (vn == ve) ? -1 : col_of(vn);
                                if (cnext < 0) { status = QPN_FAILURE; dcode = CODE_STOP; }
                            }
                        }
                    }
                }
                if (l == 0) { sDecI[0] = dcode; sDecI[1] = r; sDecI[2] = c; sDecI[3] = cnext; sDecI[4] = par; sDecD[0] = inv; }
                STAMP(6);   // (leader) first half of its turn: the decision
                __syncthreads();                            // barrier B(t): decision t is posted   (t = 0: X6)
                // ================= second half: the bookkeeping, while the H waves read the decision and publish the next column ====
                if (flip || pivot) {
                    const bool newx = c == XC;
                    const double eloW = elo, ehiW = ehi;            // the interval the entering variable lives in once basic
                    double delta, vx = 0.0, tcz0 = tcol0, tcz1 = tcol1, enter_val = 0.0, nbW = 0.0;
                    int rW = -1, kW = -1, auW = 0, vlW = ve;
                    if (flip) {
                        delta = sneg ? -self_lim : self_lim;
                        if (ve != VTH) {
                            const int au = sneg ? 0 : 1;
                            nbW = udbl(au ? sHi[kf] : sLo[kf]);
                            kW = kf; auW = au;
                            sneg = au != 0;
                            self_lim = QINF;
                            if (au) { elo = -QINF; ehi = 0.0; } else { elo = 0.0; ehi = QINF; }
                        }
                    } else {
                        if (step < 0.0) step = 0.0;
                        delta = sneg ? -step : step;
                        enter_val = udbl(sNb[c]) + delta;
                        const int rh = r >> 6, rl = r & 63;
                        // the extra column (scalar statement: prow = T[r][.] * inv; T[i][.] = fma(-cm_i, prow, T[i][.]); row r: -prow;
                        // the pivot column's own entries start from 0 and its slot in the row carries -inv)
                        if (newx) { vx = -inv; tcz0 = 0.0; tcz1 = 0.0; }
                        else vx = (rh ? readlane_f64(tcol1, rl) : readlane_f64(tcol0, rl)) * inv;
                        rW = r; vlW = vl; nbW = leave_val;
                        if (vl != VTH) {
                            const int k = vl < NBP ? vl : vl - NBP;
                            const double lk0 = udbl(sLo[k]), hk0 = udbl(sHi[k]);
                            const int cls = (lk0 == -QINF && hk0 == QINF) ? 2 : 0;
                            if (vl < NBP) {
                                // the bounded variable p_k left at a bound -- the upper one iff row r was a `hi` ratio --: d_k enters
                                const int au = uni(__double2hiint(rcr)) >= 0 ? 1 : 0;
                                kW = k; auW = au;
                                sneg = au != 0;
                                self_lim = QINF;
                                if (cls == 2) { elo = 0.0; ehi = 0.0; }
                                else if (au) { elo = -QINF; ehi = 0.0; }
                                else { elo = 0.0; ehi = QINF; }
                            } else {
                                // the multiplier d_k left at 0: p_k enters, moving off the bound it rests at
                                const int au = uni(sSat[k]);
                                sneg = au != 0;
                                self_lim = udbl(hk0 - lk0);         // +inf for a free pair
                                if (cls == 2) sneg = false;
                                elo = lk0; ehi = hk0;
                            }
                        }
                    }
                    // values, the extra column, write-backs (one definition of every vector per iteration)
                    const bool isr0 = l == rW, isr1 = 64 + l == rW;
                    xb0 = isr0 ? enter_val : fma(delta, cm0, xb0);
                    xb1 = isr1 ? enter_val : fma(delta, cm1, xb1);
                    tcol0 = isr0 ? -vx : fma(-cm0, vx, tcz0);
                    tcol1 = isr1 ? -vx : fma(-cm1, vx, tcz1);
                    if (isr0) rowvar0 = ve;
                    if (isr1) rowvar1 = ve;
                    if (!newx && l == c) colvar0 = vlW;
                    if (!newx && 64 + l == c) colvar1 = vlW;
                    if (newx) cvx = vlW;
                    if (l == 0) {
                        // the tables: row r now holds the entering variable (its interval was chosen when it was picked), column c
                        // the leaving one at the value it left at, pair k its new bound flag
                        if (rW >= 0) { sRowLo[rW] = eloW; sRowHi[rW] = ehiW; }
                        sNb[c] = nbW;
                        if (kW >= 0) sSat[kW] = auW;
                    }
                }
                code = dcode;
                if (code == CODE_STOP) break;
                __syncthreads();                            // barrier A'(t): the next column, as it stands, is in colN
                // the next column after the exchange of pivot t, derived here: T'[i][j] = T[i][j] - u_i pv_j, row r: -pv_j (a flip:
                // pv = 0; the extra column: the leader's own, already updated above)
                if (cnext == XC) { cm0 = tcol0; cm1 = tcol1; }
                else {
                    const double *const cn = colN + (l & 3) * CLD + (l >> 2);
                    const double o0 = act0 ? cn[0] : 0.0, o1 = act1 ? cn[16] : 0.0;
                    double pvn = 0.0;
                    if (code == CODE_PIVOT) pvn = udbl(((r >> 6) ? readlane_f64(o1, r & 63) : readlane_f64(o0, r & 63)) * inv);
                    const double n0 = fma(-cm0, pvn, o0), n1 = fma(-cm1, pvn, o1);
                    const bool pr = code == CODE_PIVOT;
                    cm0 = (pr && l == r) ? -pvn : n0;
                    cm1 = (pr && 64 + l == r) ? -pvn : n1;
                }
                c = cnext;
                par ^= 1;
            }
            __syncthreads();                                // every wave is out of the loop
            // the leader posts the values by variable id (the padded columns of W~ get a finite 0: 0 x 0)
            if (l < m) { sval[rowvar0] = xb0; sval[colvar0] = sNb[l]; } else sval[NBP + l] = 0.0;
            if (64 + l < m) { sval[rowvar1] = xb1; sval[colvar1] = sNb[64 + l]; } else sval[NBP + 64 + l] = 0.0;
            if (l == 0) sval[cvx] = sNb[XC];
            __syncthreads();
        } else {
            // ---- the other C waves sit the pivot loop out, barrier for barrier (B, A' per pivot), their W~ tiles untouched
            __syncthreads();                                // B(0)
            code = uni(sDecI[0]);
            while (code != CODE_STOP) {
                __syncthreads();                            // A'
                __syncthreads();                            // B
                code = uni(sDecI[0]);
            }
            __syncthreads();                                // every wave is out of the loop
            __syncthreads();                                // the leader has posted the values by variable id
        }
        // ... and then forms its rows of x = W~ lambda - h: (W~ lambda)_row for the 4 rows (g) a lane holds, folded butterfly
        // over the 16 lanes of a DPP row (see the 64-class)
        l = lane_id_fresh(); lc = l & 15; lq = l >> 4;
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#define M_ROWP(J, T)                                                                                \
    if constexpr ((J) < NR) {                                                                       \
        const double lj = sval[NBP + 16 * (J) + lc];                                                \
        p0 = fma(T[0], lj, p0); p1 = fma(T[1], lj, p1); p2 = fma(T[2], lj, p2); p3 = fma(T[3], lj, p3); \
    }
        M_ROWP(0, tc0) M_ROWP(1, tc1) M_ROWP(2, tc2) M_ROWP(3, tc3) M_ROWP(4, tc4) M_ROWP(5, tc5) M_ROWP(6, tc6) M_ROWP(7, tc7)
#undef M_ROWP
        const bool b0 = (lc & 1) != 0, b1 = (lc & 2) != 0;
        const double q0 = (b0 ? p1 : p0) + dpp_f64<0xB1>(b0 ? p0 : p1), q1 = (b0 ? p3 : p2) + dpp_f64<0xB1>(b0 ? p2 : p3);
        double t0 = (b1 ? q1 : q0) + dpp_f64<0x4E>(b1 ? q0 : q1);       // g = b0 + 2 b1, summed over lc bits 0, 1
        t0 += __shfl_xor(t0, 4, 64);
        t0 += __shfl_xor(t0, 8, 64);
        if (lc < 4) {
            const int row = 16 * I + 4 * lc + lq;                       // g = lc for lc < 4
            sz[row] = t0 - sH[row];
        }
        }
    }
#undef M_STAGE_AD
#undef M_SCALE
#undef M_TILE_OF
#undef M_PANEL
#undef M_STEP4
#undef M_ELIM_C
#undef M_ELIM_H
#undef M_UPRIME
#undef M_FACTOR
#undef M_PUB_C
#undef M_PUB_H
#undef M_PUBT
#undef M_PUBC
#undef M_UPD
#undef M_UPDC
    (void)any_decline;
    if (fail) { decline(); return; }        // an equality row, or a block pivot below the threshold: the general path takes the node

    // ---- the tail: z in item order, post-check (everything below derives its lane coordinates and kernel arguments afresh)
    l = lane_id_fresh(); lc = l & 15; lq = l >> 4; tid = 64 * v + l;
    kargs_t kp = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
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
    // Four lanes per row: the workgroup has 128 NR threads for at most 32 NR rows, and with one workgroup per CU nothing else
    // hides a 2 x 128-term chain and its loads.  Lanes lc, lc + 16, lc + 32, lc + 48 of wave v take a quarter each of row
    // 16 v + lc of [[Qd, -Ad'],[Ad, 0]] z; all of a lane's entries of Qd (at most 32) are requested at once; `xsum_rows` adds the
    // quarters.  Finite blocks: a zero z_j contributes exactly nothing.
    const int k = 16 * v + lc;
    const bool rowok = k < N;
    const int gk = k >= n;
    double part;
    {
        const int jq = (n + 3) >> 2, jlo = lq * jq, jcnt = jlo < n ? (n - jlo < jq ? n - jlo : jq) : 0;      // this quarter's columns of x
        double pa = 0.0, pb = 0.0;
        if (rowok && !gk) {
            double mv[32];
            const int jb = jcnt > 0 ? jlo : 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) mv[q] = Qe_[(size_t)(q < jcnt ? jlo + q : jb) * n + k];
#pragma unroll
            for (int q = 0; q < 32; q += 2) {
                pa = (q < jcnt) ? fma(mv[q], sz[jb + q], pa) : pa;
                pb = (q + 1 < jcnt) ? fma(mv[q + 1], sz[jb + q + 1], pb) : pb;
            }
            const int iq = (m + 3) >> 2, ilo = lq * iq, ihi = ilo + iq < m ? ilo + iq : m;                  // ... and of lambda: -Ad'
            int i = ilo;
            for (; i + 2 <= ihi; i += 2) { pa = fma(-sAd[k * LDA + i], sz[n + i], pa); pb = fma(-sAd[k * LDA + i + 1], sz[n + i + 1], pb); }
            if (i < ihi) pa = fma(-sAd[k * LDA + i], sz[n + i], pa);
        } else if (rowok) {
            const int r = k - n, jhi = jlo + jcnt;
            int j = jlo;
            for (; j + 2 <= jhi; j += 2) { pa = fma(sAd[j * LDA + r], sz[j], pa); pb = fma(sAd[(j + 1) * LDA + r], sz[j + 1], pb); }
            if (j < jhi) pa = fma(sAd[j * LDA + r], sz[j], pa);
        }
        part = xsum_rows(pa + pb);                          // (every lane takes part)
    }
    if (rowok && lq == 0) {
        const double rk = sQ[k] + part;
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
#ifdef QPN_STAMPS
    if (tid == 0 && kp->stamps) {                       // H wave 0's phases: the second half of a [2][batch][8] buffer
        STAMP(0);
        for (int k = 0; k < 8; ++k) kp->stamps[((size_t)kp->batch + b) * 8 + k] = stamp_acc[k];
    }
#endif
    if (tid == 64 * NR) {                               // the leader's lane 0: it holds the status and the pivot count
        if (badt > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
#ifdef QPN_STAMPS
        STAMP(0);   // (diagnostic builds: read-back + post-check are added to the load slot)
        if (kp->stamps) for (int k = 0; k < 8; ++k) kp->stamps[(size_t)b * 8 + k] = stamp_acc[k];
#endif
        kp->status[b] = status;
        double rs = sRed[0];
#pragma unroll
        for (int k = 1; k < 2 * NR; ++k) rs = sRed[k] > rs ? sRed[k] : rs;
        if (re_) re_[b] = rs;
        if (pe_) pe_[b] = pivots;
        if (int32_t *const sk = kp->sched_key) { const int k0 = sk[b]; sk[b] = k0 > 0 ? k0 - (k0 >> 5) + pivots : 32 * pivots; }      // smoothed pivot count
    }
}

} // namespace

bool qpn_schur_wg2_shape(int n, int m)
{
    return (n > 64 || m > 64) && n >= 1 && n <= 128 && m >= 1 && m <= 128;
}

// One launch: every node of the batch solved, checked and written back by its own workgroup.  Nodes the kernel declines
// keep status -1.
hipError_t qpn_launch_schur_wg2_nodes(const AviBatchArgs &a, hipStream_t stream)
{
    const int n = a.nd.n, m = a.nd.m, batch = a.batch;
    if (batch <= 0) return hipSuccess;
    const int big = n > m ? n : m;
    const int NR = big <= 80 ? 5 : big <= 96 ? 6 : big <= 112 ? 7 : 8;
    const size_t lds = (size_t)wg2_lds_doubles(n, m, 16 * NR) * sizeof(double);
    static QpnPerDeviceOnce once;
    const int dev = once.device();
    if (!once.done[dev]) {
        // (dynamic LDS beyond 64 KB needs the attribute, once per device and kernel; the largest class needs 151 KB)
        const int mx = 156 * 1024;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_wg2_nodes<5>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_wg2_nodes<6>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_wg2_nodes<7>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_wg2_nodes<8>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e != hipSuccess) return e;
        once.done[dev] = true;
    }
    const dim3 grid((unsigned)batch);
    switch (NR) {
    case 5: hipLaunchKernelGGL((schur_wg2_nodes<5>), grid, dim3(640), lds, stream, a); break;
    case 6: hipLaunchKernelGGL((schur_wg2_nodes<6>), grid, dim3(768), lds, stream, a); break;
    case 7: hipLaunchKernelGGL((schur_wg2_nodes<7>), grid, dim3(896), lds, stream, a); break;
    default: hipLaunchKernelGGL((schur_wg2_nodes<8>), grid, dim3(1024), lds, stream, a); break;
    }
    return hipGetLastError();
}
