// qpn_avi_reg.hip -- register-resident batched box-MCP / GAVI pivotal solver for gfx950 (CDNA4).
//
// Replaces PATHSolver.solve_mcp as called at src/avi.jl:64-70 / src/qp_processing.jl:22-27 (+ the
// post-check of src/avi.jl:71-76, :148-156 and comp_indices, src/avi_solutions.jl:511-562) for
// batches of independent node-AVIs, N <= 64.  Same algorithm and the same arithmetic, bit for
// bit, as the CPU checker used by the tests -- see DESIGN.md section 3.
//
// Machine mapping.  ONE 64-lane wavefront per AVI.  The lanes form an 8 x 8 grid; lane (ra, cb)
// keeps the BS x BS block T[BS*ra.., BS*cb..] of the dictionary in VGPRs (BS = 8 for N <= 64:
// 128 VGPRs, 2 waves/SIMD).  A pivot is the rank-1 update T -= u v': a lane needs only BS entries
// of u (pivot column) and BS of v (scaled pivot row), fetched from two padded LDS vectors with
// conflict-free ds_read_b128 -- BS*BS v_fma_f64 and ~16 LDS instructions per lane per pivot.
// Row/column vectors (basic values, covering column, cached admissible intervals, bookkeeping)
// live one element per lane.  All control is wave-uniform: pivot row/column ids are SGPRs
// (ballot/ffs, v_readlane), reductions run on DPP row operations + v_readlane (no LDS round trips),
// and the dynamically selected pivot row/column are read/written IN PLACE by the 8 owner lanes
// behind a scalar binary dispatch on the block-local index (no dynamic VGPR indexing, no copies).
// HBM: the stacked M block is read with coalesced 512-byte column loads, BS columns at a time,
// transposed through a 5 KB LDS stage; the post-check re-reads it (served by L2 / Infinity Cache).
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

// Every workgroup of this file is ONE wavefront: LDS operations of a wave execute in issue order, so ordering between a
// lane's store and another lane's load needs no s_barrier and no drain of the memory counters -- only that the compiler
// keeps the program order of the LDS accesses and does not move them across this point (as in qpn_avi_schur.hip).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int BS> struct Geo {
    static constexpr int NB = 8 * BS;        // padded dimension held in registers
    static constexpr int PB = BS + 2;        // padded block stride (doubles) in the LDS vectors
    static constexpr int NP = 8 * PB;        // padded vector length
    __device__ static __forceinline__ int pidx(int i) { return (i / BS) * PB + (i % BS); }
};

// Scalar binary dispatch on a wave-uniform block-local index sel in [0, 8): three s_cmp/s_cbranch
// levels, each leaf names its registers statically.  (Written as a macro: lambdas capturing the
// register block by reference make hipcc keep the block in scratch.)
#define QPN_DISPATCH8(sel, LEAF)                                                               \
    do {                                                                                       \
        if ((sel) < 4) {                                                                       \
            if ((sel) < 2) { if ((sel) < 1) { LEAF(0) } else { LEAF(1) } }                     \
            else { if ((sel) < 3) { LEAF(2) } else { LEAF(3) } }                               \
        } else {                                                                               \
            if ((sel) < 6) { if ((sel) < 5) { LEAF(4) } else { LEAF(5) } }                     \
            else { if ((sel) < 7) { LEAF(6) } else { LEAF(7) } }                               \
        }                                                                                      \
    } while (0)

// The BS x BS register block is 64 individually named scalars (E(k,l)), never an array: any
// array form (even fully unrolled) ends up address-taken somewhere and hipcc parks it in scratch.
#define E(k, l) t_##k##_##l
#define QPN_FOR_L(M, k) M(k, 0) M(k, 1) M(k, 2) M(k, 3) M(k, 4) M(k, 5) M(k, 6) M(k, 7)
#define QPN_FOR_K(M, l) M(0, l) M(1, l) M(2, l) M(3, l) M(4, l) M(5, l) M(6, l) M(7, l)
#define QPN_FOR_KL(M)                                                                          \
    QPN_FOR_L(M, 0) QPN_FOR_L(M, 1) QPN_FOR_L(M, 2) QPN_FOR_L(M, 3)                            \
    QPN_FOR_L(M, 4) QPN_FOR_L(M, 5) QPN_FOR_L(M, 6) QPN_FOR_L(M, 7)
#define QPN_FOR_1(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

// admissible interval of variable v (wave-uniform id) while it is basic
__device__ __forceinline__ void interval_uni(int v, int N, const double *sl, const double *su,
                                             const int *sat, double &lo, double &hi)
{
    if (v == 2 * N) { lo = 0.0; hi = QINF; return; }
    if (v < N) { lo = udbl(sl[v]); hi = udbl(su[v]); return; }
    const int k = v - N;
    const double L = udbl(sl[k]), U = udbl(su[k]);
    if (L == U) { lo = -QINF; hi = QINF; }
    else if (L == -QINF && U == QINF) { lo = 0.0; hi = 0.0; }
    else if (uni(sat[k])) { lo = -QINF; hi = 0.0; }
    else { lo = 0.0; hi = QINF; }
}

// One item, one wavefront (the whole kernel body; see avi_solve_reg below for how items are picked).
// LEAN: the reduced problems of the mid-size node path (QPN_AVI_IFLAG_BLOCKED_M: every row a GAVI row with l < u, cold start,
// M in the register-block layout) -- no crash stage, no staged load, no post-check: compiled out, not branched around
template <int BS, bool LEAN = false>
__device__ __forceinline__ void avi_solve_reg_item(const AviBatchArgs &a, const int b)
{
    using G = Geo<BS>;
    constexpr int NB = G::NB, PB = G::PB, NP = G::NP;
    constexpr int XC = NB;                 // index of the extra (covering) column
    const int N = a.N;
    const int lane = threadIdx.x;
    const int ra = lane >> 3, cb = lane & 7;
    const bool act = lane < N;             // this lane carries row `lane` / column `lane`

    __shared__ __attribute__((aligned(16))) double ucol[NP + 2];
    __shared__ __attribute__((aligned(16))) double vrow[NP + 2];   // vrow[NP] = extra-column entry
    __shared__ __attribute__((aligned(16))) double stage[BS * NP];
    __shared__ double sl[NB], su[NB], snb[2 * NB + 2];
    __shared__ int sat[NB], elist[8 * NB + 8];

#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    const double *Mg = a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)(a.vec_stride ? a.vec_stride : N);     // vectors of item b (reduced problems keep the parent's stride)
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;

    // ---- pair k = lane ----------------------------------------------------------------
    int gk = (act && a.kind) ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + lane] : 0;
    bool freek, fixedk;
    int atup0 = 0;
    double v0 = 0.0;
    int rowvar = -1, colvar = -1;
    double lo = -QINF, hi = QINF;   // admissible interval of this row's basic variable (kept current)
    int cNvar = 2 * N;
    double cNval = 0.0;
    double tcol = 0.0;                      // extra column, one entry per row-lane
    double xb = act ? a.q[vo + lane] : 0.0;
    {
    const double lk = act ? a.l[vo + lane] : 0.0;
    const double uk = act ? a.u[vo + lane] : 0.0;
    freek = act && lk == -QINF && uk == QINF;
    fixedk = act && lk == uk;
    if (act) {
        if (gk) { colvar = N + lane; rowvar = lane; }
        else {
            double z0 = (a.flags & QPN_AVI_FLAG_COLD_START) ? 0.0 : a.z[vo + lane];
            if (isnan(z0)) z0 = 0.0;
            if (freek) v0 = z0;
            else {
                if (z0 < lk) z0 = lk;
                if (z0 > uk) z0 = uk;
                if (lk == -QINF) { v0 = uk; atup0 = 1; }
                else if (uk == QINF) { v0 = lk; }
                else if (uk - z0 < z0 - lk) { v0 = uk; atup0 = 1; }
                else v0 = lk;
                if (uk == lk) atup0 = 0;
            }
            colvar = lane; rowvar = N + lane;
        }
        sl[lane] = lk; su[lane] = uk; sat[lane] = atup0; snb[lane] = v0;
    }
    if (act) {
        if (gk) { lo = lk; hi = uk; }                       // p_k = (Mz+q)_k basic
        else if (fixedk) { lo = -QINF; hi = QINF; }         // d_k of a fixed pair: free
        else if (freek) { lo = 0.0; hi = 0.0; }             // equation row
        else if (atup0) { lo = -QINF; hi = 0.0; }
        else { lo = 0.0; hi = QINF; }
    }
    }
    double nbval = v0;
    wave_sync();

    // ---- load M: coalesced HBM -> LDS stage (BS columns at a time) -> register blocks ------
#define M_DECL(k, l) double E(k, l) = 0.0;
    QPN_FOR_KL(M_DECL)
#undef M_DECL
    // Software-pipelined, two column blocks deep: the coalesced (512 B per instruction) loads of
    // blocks k+1..k+2 are in flight while block k is transposed through the LDS stage (an HBM
    // round trip under load is several thousand cycles; one block of transposition is ~2 K).
    // Element idx of a block lives at (col, row) = (idx / N, idx % N); a lane fetches
    // idx = lane + 64 s, s < BS.  The two buffers are named scalars (static rotation).
    double mabs = 0.0;                       // running max |M_ij| of the elements this lane moves
#define PF(B, sidx) pf##B##_##sidx
#define M_PFDECL(sidx) double PF(a, sidx) = 0.0, PF(b, sidx) = 0.0;
    QPN_FOR_1(M_PFDECL)
#undef M_PFDECL
#define QPN_ISSUE(B, cbk)                                                                       \
    {                                                                                           \
        const int col0_ = BS * (cbk);                                                           \
        const int nc_ = (N - col0_) < BS ? (N - col0_) : BS;                                    \
        const int cnt_ = col0_ < N ? nc_ * N : 0;                                               \
        const double *base_ = Mg + (size_t)col0_ * N;                                           \
        if constexpr (0 < BS) PF(B, 0) = (lane + WAVE * 0 < cnt_) ? base_[lane + WAVE * 0] : 0.0; \
        if constexpr (1 < BS) PF(B, 1) = (lane + WAVE * 1 < cnt_) ? base_[lane + WAVE * 1] : 0.0; \
        if constexpr (2 < BS) PF(B, 2) = (lane + WAVE * 2 < cnt_) ? base_[lane + WAVE * 2] : 0.0; \
        if constexpr (3 < BS) PF(B, 3) = (lane + WAVE * 3 < cnt_) ? base_[lane + WAVE * 3] : 0.0; \
        if constexpr (4 < BS) PF(B, 4) = (lane + WAVE * 4 < cnt_) ? base_[lane + WAVE * 4] : 0.0; \
        if constexpr (5 < BS) PF(B, 5) = (lane + WAVE * 5 < cnt_) ? base_[lane + WAVE * 5] : 0.0; \
        if constexpr (6 < BS) PF(B, 6) = (lane + WAVE * 6 < cnt_) ? base_[lane + WAVE * 6] : 0.0; \
        if constexpr (7 < BS) PF(B, 7) = (lane + WAVE * 7 < cnt_) ? base_[lane + WAVE * 7] : 0.0; \
    }
#define QPN_PUT(B, sidx)                                                                        \
    if constexpr ((sidx) < BS) {                                                                \
        if (lane + WAVE * (sidx) < cnt) {                                                       \
            stage[col * NP + G::pidx(row)] = PF(B, sidx);                                       \
            mabs = fmax(mabs, fabs(PF(B, sidx)));                                               \
        }                                                                                       \
        row += WAVE;                                                                            \
        while (row >= N) { row -= N; col++; }                                                   \
    }
#define QPN_PUT_FULL(B, sidx)   /* N == 64: element (col, row) = (sidx, lane) */                 \
    if constexpr ((sidx) < BS) {                                                                \
        stage[(sidx) * NP + G::pidx(lane)] = PF(B, sidx);                                       \
        mabs = fmax(mabs, fabs(PF(B, sidx)));                                                   \
    }
#define QPN_ROUND(B, cbk)                                                                       \
    if (BS * (cbk) < N) {                                                                       \
        const int col0 = BS * (cbk);                                                            \
        const int ncols = (N - col0) < BS ? (N - col0) : BS;                                    \
        const int cnt = ncols * N;                                                              \
        if (full) {                                                                             \
            QPN_PUT_FULL(B, 0) QPN_PUT_FULL(B, 1) QPN_PUT_FULL(B, 2) QPN_PUT_FULL(B, 3)         \
            QPN_PUT_FULL(B, 4) QPN_PUT_FULL(B, 5) QPN_PUT_FULL(B, 6) QPN_PUT_FULL(B, 7)         \
        } else {                                                                                \
            int row = lane, col = 0;                                                            \
            while (row >= N) { row -= N; col++; }                                               \
            QPN_PUT(B, 0) QPN_PUT(B, 1) QPN_PUT(B, 2) QPN_PUT(B, 3)                             \
            QPN_PUT(B, 4) QPN_PUT(B, 5) QPN_PUT(B, 6) QPN_PUT(B, 7)                             \
        }                                                                                       \
        if ((cbk) + 2 < 8) QPN_ISSUE(B, (cbk) + 2)                                              \
        wave_sync();                                                                        \
        /* initial basic values xb = q + M z_nb, columns in ascending order (as the checker) */ \
        for (int l = 0; l < ncols; ++l) {                                                       \
            const double zj = snb[col0 + l];                                                    \
            if (zj != 0.0 && act) xb = fma(stage[l * NP + G::pidx(lane)], zj, xb);              \
        }                                                                                       \
        if (cb == (cbk)) { if (full) { QPN_FOR_KL(M_LOAD_FULL) } else { QPN_FOR_KL(M_LOAD) } }  \
        wave_sync();                                                                        \
    }
#define M_LOAD(k, l)                                                                            \
    if constexpr ((k) < BS && (l) < BS)                                                         \
        E(k, l) = (BS * ra + (k) < N && (l) < ncols) ? stage[(l) * NP + ra * PB + (k)] : 0.0;
#define M_LOAD_FULL(k, l)                                                                       \
    if constexpr ((k) < BS && (l) < BS) E(k, l) = stage[(l) * NP + ra * PB + (k)];
    const bool full = N == WAVE && BS == 8;
    const bool blocked = LEAN || (a.flags & QPN_AVI_IFLAG_BLOCKED_M) != 0;        // wave-uniform
    if (blocked) {
        // the caller wrote M in this kernel's own register-block layout, lane fastest: entry (k, l) of lane's block at
        // [(l BS + k) 64 + lane] -- every load instruction reads 512 consecutive bytes
        const double *Sb = Mg + lane;
#define M_LOADB(k, l)                                                                           \
    if constexpr ((k) < BS && (l) < BS) {                                                       \
        const double t_ = Sb[((l) * BS + (k)) * 64];                                            \
        E(k, l) = (BS * ra + (k) < N && BS * cb + (l) < N) ? t_ : 0.0;                          \
    }
        QPN_FOR_KL(M_LOADB)
#undef M_LOADB
        mabs = 1.0;
    } else {
    QPN_ISSUE(a, 0)
    QPN_ISSUE(b, 1)
    QPN_ROUND(a, 0) QPN_ROUND(b, 1) QPN_ROUND(a, 2) QPN_ROUND(b, 3)
    QPN_ROUND(a, 4) QPN_ROUND(b, 5) QPN_ROUND(a, 6) QPN_ROUND(b, 7)
    }
#undef M_LOAD_FULL
#undef QPN_PUT_FULL
#undef M_LOAD
#undef QPN_ROUND
#undef QPN_PUT
#undef QPN_ISSUE
#undef PF
    STAMP(0);   // setup + load
    int pivots = 0;
    int pivots_init = 0;

    auto col_of = [&](int v) -> int {
        int c = wave_first(act && colvar == v);
        if (c < 0 && cNvar == v) c = XC;
        return c;
    };

    // ---- one pivot loop for both stages (a single instance of the update code) --------------
    //   stage 0: crash -- free variables / multipliers of equality GAVI rows enter (Stage A)
    //   stage 1: build the covering column from the basic infeasibilities
    //   stage 2: Lemke's complementary pivoting (Stage B)
    int n_enter;
    {
        bool want = act && ((!gk && freek) || (gk && fixedk));
        unsigned long long bm = __ballot(want);
        n_enter = __popcll(bm);
        if (want) {
            int pos = __popcll(bm & ((1ull << lane) - 1ull));
            elist[pos] = gk ? N + lane : lane;
        }
    }
    wave_sync();
    int budget = 4 * N + 4;
    int stage_ = 0, idx = 0;
    int status = QPN_FAILURE;

    // ---- Stage A fast path: diagonal crash pivots, statically unrolled --------------------------
    // While the next variable to enter is a free z_e whose column and equation row are still its
    // own and |T[e][e]| >= 1e-4 max(1, max|M|), the pivot is (e, e): no search, no ratio test, and
    // the block-local indices are compile-time constants (no dispatch).  Same arithmetic as the
    // general loop below, which takes over at the first variable that does not qualify.
    const double mscale = wave_max_f64(mabs);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);
    bool fast_ok = !LEAN;
    int pivots_blk = 0;
    // ---- blocked variant (BS = 8): four consecutive diagonal crash pivots e0..e0+3 per step -----
    // The 4 pivot columns / rows are copied to LDS once; the 64 row/column lanes run the sequential
    // elimination on that 64x4 + 4x64 panel in registers (same fma sequence as four single pivots),
    // publish the four (u_k, v_k) pairs, and every lane applies the four rank-1 updates to its
    // register block back to back: one extraction and one panel per 4 pivots instead of 4 round
    // trips.  Bit-identical to four single pivots.
    if constexpr (BS == 8) {
#define M_B_XC(k, L) stage[(L & 3) * NP + ra * PB + (k)] = E(k, L);
#define M_B_XR(K, l) stage[(4 + (K & 3)) * NP + cb * PB + (l)] = E(K, l);
#define M_B_LDU(k) const double bu_##k = stage[kq * NP + ra * PB + (k)];
#define M_B_LDV(l) const double bv_##l = stage[(4 + kq) * NP + cb * PB + (l)];
#define M_B_FMA(k, l) E(k, l) = fma(-bu_##k, bv_##l, E(k, l));
#define M_B_FC(k, L) E(k, L) = bu_##k * binv;
#define M_B_FR(K, l) E(K, l) = -bv_##l;
#define M_B_APPLY(KQ, L0)                                                                          \
    if (nk > (KQ)) {                                                                               \
        constexpr int kq = (KQ);                                                                   \
        const double binv = (KQ) == 0 ? inv0 : (KQ) == 1 ? inv1 : (KQ) == 2 ? inv2 : inv3;         \
        QPN_FOR_1(M_B_LDU)                                                                         \
        QPN_FOR_1(M_B_LDV)                                                                         \
        QPN_FOR_KL(M_B_FMA)                                                                        \
        if (cb == jbk) { QPN_FOR_K(M_B_FC, L0) }                                                   \
        if (ra == jbk) { QPN_FOR_L(M_B_FR, L0) }                                                   \
    }
#define M_B_PANEL(KQ)                                                                              \
    if (nk == (KQ)) {                                                                              \
        const int ek = e0 + (KQ);                                                                  \
        const double piv = readlane_f64(cp##KQ, ek);                                               \
        if (ubool(fabs(piv) >= diag_thr)) {                                                        \
            const double inv = 1.0 / piv;                                                          \
            const double delta = udbl((0.0 - readlane_f64(xb, ek)) * inv);                         \
            const double ent = udbl(readlane_f64(nbval, ek) + delta);                              \
            const double uk = cp##KQ;                                                              \
            double vk = rp##KQ * inv;                                                              \
            xb = fma(delta, uk, xb);                                                               \
            if (lane == ek) { xb = ent; rowvar = ek; colvar = N + ek; nbval = 0.0; lo = -QINF; hi = QINF; vk = -inv; } \
            if ((KQ) == 0) inv0 = inv; else if ((KQ) == 1) inv1 = inv; else if ((KQ) == 2) inv2 = inv; else inv3 = inv; \
            /* remaining panel columns / rows see pivot KQ exactly as the full update would */    \
            if ((KQ) < 1) { const double vke = readlane_f64(vk, e0 + 1), uke = readlane_f64(uk, e0 + 1); \
                cp1 = (lane == ek) ? -vke : fma(-uk, vke, cp1); rp1 = (lane == ek) ? uke * inv : fma(-uke, vk, rp1); } \
            if ((KQ) < 2) { const double vke = readlane_f64(vk, e0 + 2), uke = readlane_f64(uk, e0 + 2); \
                cp2 = (lane == ek) ? -vke : fma(-uk, vke, cp2); rp2 = (lane == ek) ? uke * inv : fma(-uke, vk, rp2); } \
            if ((KQ) < 3) { const double vke = readlane_f64(vk, e0 + 3), uke = readlane_f64(uk, e0 + 3); \
                cp3 = (lane == ek) ? -vke : fma(-uk, vke, cp3); rp3 = (lane == ek) ? uke * inv : fma(-uke, vk, rp3); } \
            /* publish (u_k, v_k) over the panel slots (all lanes have read them already) */      \
            stage[(KQ) * NP + G::pidx(lane)] = uk;                                                 \
            stage[(4 + (KQ)) * NP + G::pidx(lane)] = vk;                                           \
            nk++;                                                                                  \
        }                                                                                          \
    }
#define M_BLK(H, L0, L1, L2, L3)                                                                    \
    {                                                                                               \
        const int e0 = 8 * jbk + 4 * (H);                                                           \
        bool go = fast_ok && idx + 3 < n_enter && budget >= 4 && e0 + 3 < N;                        \
        if (go) {                                                                                   \
            go = uni(elist[idx]) == e0 && uni(elist[idx + 1]) == e0 + 1 &&                          \
                 uni(elist[idx + 2]) == e0 + 2 && uni(elist[idx + 3]) == e0 + 3;                    \
            const unsigned long long ownm = __ballot(lane >= e0 && lane < e0 + 4 && colvar == lane && rowvar == N + lane); \
            go = go && __popcll(ownm) == 4;                                                         \
        }                                                                                           \
        if (go) {                                                                                   \
            if (cb == jbk) { QPN_FOR_K(M_B_XC, L0) QPN_FOR_K(M_B_XC, L1)                            \
                             QPN_FOR_K(M_B_XC, L2) QPN_FOR_K(M_B_XC, L3) }                          \
            if (ra == jbk) { QPN_FOR_L(M_B_XR, L0) QPN_FOR_L(M_B_XR, L1)                            \
                             QPN_FOR_L(M_B_XR, L2) QPN_FOR_L(M_B_XR, L3) }                          \
            wave_sync();                                                                        \
            double cp0 = stage[0 * NP + G::pidx(lane)], cp1 = stage[1 * NP + G::pidx(lane)];        \
            double cp2 = stage[2 * NP + G::pidx(lane)], cp3 = stage[3 * NP + G::pidx(lane)];        \
            double rp0 = stage[4 * NP + G::pidx(lane)], rp1 = stage[5 * NP + G::pidx(lane)];        \
            double rp2 = stage[6 * NP + G::pidx(lane)], rp3 = stage[7 * NP + G::pidx(lane)];        \
            wave_sync();                                                                        \
            double inv0 = 0.0, inv1 = 0.0, inv2 = 0.0, inv3 = 0.0;                                  \
            int nk = 0;                                                                             \
            M_B_PANEL(0) M_B_PANEL(1) M_B_PANEL(2) M_B_PANEL(3)                                     \
            (void)cp0; (void)rp0;                                                                   \
            wave_sync();                                                                        \
            M_B_APPLY(0, L0) M_B_APPLY(1, L1)                                                       \
            M_B_APPLY(2, L2) M_B_APPLY(3, L3)                                                       \
            pivots_blk += nk; idx += nk; budget -= nk;                                              \
            if (nk < 4) fast_ok = false;                                                            \
            wave_sync();                                                                        \
        }                                                                                           \
    }
        for (int jbk = 0; jbk < 8 && fast_ok; ++jbk) {
            M_BLK(0, 0, 1, 2, 3)
            M_BLK(1, 4, 5, 6, 7)
            // a block that did not qualify as a whole is left to the single-pivot path below
            if (!(idx < n_enter) || uni(elist[idx < n_enter ? idx : 0]) != 8 * (jbk + 1)) break;
        }
#undef M_BLK
#undef M_B_PANEL
#undef M_B_APPLY
#undef M_B_FR
#undef M_B_FC
#undef M_B_FMA
#undef M_B_LDV
#undef M_B_LDU
#undef M_B_XR
#undef M_B_XC
    }
    {
#define M_FAST(JB, JJ)                                                                              \
    if constexpr ((JJ) < BS) {                                                                      \
        const int e_ = BS * (JB) + (JJ);                                                            \
        if (fast_ok && idx < n_enter && budget > 0 && uni(elist[idx < n_enter ? idx : 0]) == e_ &&  \
            e_ < N) {                                                                               \
            const bool own = readlane_i32(colvar, e_) == e_ && readlane_i32(rowvar, e_) == N + e_;  \
            if (!own) { fast_ok = false; }                                                          \
            else {                                                                                  \
                if (cb == (JB)) { QPN_FOR_K(M_FXC, JJ) }                                            \
                if (ra == (JB)) { QPN_FOR_L(M_FXR, JJ) }                                            \
                wave_sync();                                                                    \
                /* one LDS round trip: pivot, this row-lane's column entry, u and raw v blocks */   \
                const double pivr = ucol[G::pidx(e_)];                                              \
                const double cmf = lane < NB ? ucol[G::pidx(lane)] : 0.0;                           \
                QPN_FOR_1(M_FLDU)                                                                   \
                QPN_FOR_1(M_FLDVR)                                                                  \
                const double piv = udbl(pivr);                                                      \
                if (!ubool(fabs(piv) >= diag_thr)) { fast_ok = false; }                             \
                else {                                                                              \
                    const double inv = 1.0 / piv;                                                   \
                    const double delta = udbl((0.0 - readlane_f64(xb, e_)) * inv);                  \
                    const double ent = udbl(readlane_f64(nbval, e_) + delta);                       \
                    xb = fma(delta, cmf, xb);                                                       \
                    if (lane == e_) { xb = ent; rowvar = e_; colvar = N + e_; nbval = 0.0; lo = -QINF; hi = QINF; } \
                    QPN_FOR_1(M_FLDV)                                                               \
                    if (cb == (JB)) v_##JJ##_f = -inv;                                              \
                    QPN_FOR_KL(M_FFMA)                                                              \
                    if (cb == (JB)) { QPN_FOR_K(M_FFC, JJ) }                                        \
                    if (ra == (JB)) { QPN_FOR_L(M_FFR, JJ) }                                        \
                    pivots_fast++; idx++; budget--;                                                 \
                    wave_sync();                                                                \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    }
#define M_FXC(k, L) if constexpr ((k) < BS) ucol[ra * PB + (k)] = E(k, L);
#define M_FXR(K, l) if constexpr ((l) < BS) vrow[cb * PB + (l)] = E(K, l);
#define M_FLDU(k) double u_##k##_f = 0.0; if constexpr ((k) < BS) u_##k##_f = ucol[ra * PB + (k)];
#define M_FLDVR(l) double vr_##l##_f = 0.0; if constexpr ((l) < BS) vr_##l##_f = vrow[cb * PB + (l)];
#define M_FLDV(l) double v_##l##_f = vr_##l##_f * inv;
#define M_FFMA(k, l) if constexpr ((k) < BS && (l) < BS) E(k, l) = fma(-u_##k##_f, v_##l##_f, E(k, l));
#define M_FFC(k, L) if constexpr ((k) < BS) E(k, L) = u_##k##_f * inv;
#define M_FFR(K, l) if constexpr ((l) < BS) E(K, l) = -v_##l##_f;
#define M_FAST_ROW(JB) M_FAST(JB, 0) M_FAST(JB, 1) M_FAST(JB, 2) M_FAST(JB, 3) M_FAST(JB, 4) M_FAST(JB, 5) M_FAST(JB, 6) M_FAST(JB, 7)
        int pivots_fast = 0;
        for (int jb = 0; jb < 8 && fast_ok; ++jb) {
            // one iteration per column block; inside, the 8 block-local indices are unrolled
            const int JBv = jb;
            (void)JBv;
#define JBX jb
            M_FAST_ROW(JBX)
#undef JBX
        }
        pivots_init = pivots_fast + pivots_blk;
        STAMP(6);   // crash fast path
#undef M_FAST_ROW
#undef M_FFR
#undef M_FFC
#undef M_FFMA
#undef M_FLDV
#undef M_FLDVR
#undef M_FLDU
#undef M_FXR
#undef M_FXC
#undef M_FAST
    }
    pivots = pivots_init;

    int c = XC;
    double sigma = -1.0, self_lim = 0.0;
    double elo = 0.0, ehi = QINF;             // admissible interval of the entering variable (stage 2)
    const double slack = 1e-10;
    const double ptol = a.piv_tol;
    for (;;) {
        int e = 0;
        if (stage_ == 0) {
            if (!(idx < n_enter && budget > 0)) { stage_ = 1; continue; }
            e = uni(elist[idx]);
            idx++;
            c = wave_first(act && colvar == e);
            if (c < 0) continue;
        } else if (stage_ == 1) {
            double viol = 0.0;
            if (act) viol = xb < lo ? lo - xb : (xb > hi ? xb - hi : 0.0);
            const double theta0 = wave_max_f64(viol);
            if (ubool(theta0 <= a.feas_tol)) { status = QPN_SUCCESS; break; }
            if (act) {
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
            cNval = theta0;
            c = XC; sigma = -1.0; self_lim = theta0;
            elo = 0.0; ehi = QINF;            // the artificial: [0, inf)
            status = QPN_MAX_ITERS;
            stage_ = 2;
            continue;
        } else {
            if (pivots >= max_piv) break;
        }
        c = uni(c);
        STAMP(1);   // loop control / stage setup
        double cm;
        {
        if (c == XC) {
            if (lane < NB) ucol[G::pidx(lane)] = tcol;
        } else if (cb == c / BS) {
            const int csel = c % BS;
#define M_XC(k, L) if constexpr ((k) < BS && (L) < BS) ucol[ra * PB + (k)] = E(k, L);
#define QPN_LEAF(L) QPN_FOR_K(M_XC, L)
            QPN_DISPATCH8(csel, QPN_LEAF);
#undef QPN_LEAF
#undef M_XC
        }
        wave_sync();
            cm = lane < NB ? ucol[G::pidx(lane)] : 0.0;
        }
        STAMP(2);   // column extraction
        int r;
        double delta, leave_val, inv;
        if (stage_ == 0) {
            const double av = act ? fabs(cm) : 0.0;
            const double colmax = wave_max_f64(av);
            const double thresh = 1e-9 * (colmax > 1.0 ? colmax : 1.0);
            r = -1;
            leave_val = 0.0;
            // diagonal first: a free variable whose column is still its own takes its own
            // equation row when the pivot reaches 1e-4 max(1, max|M|) -- no search
            if (e < N && c == e && readlane_i32(rowvar, e) == N + e) {
                const double ad = readlane_f64(av, e);
                if (ubool(ad >= diag_thr)) r = e;
            }
            if (r < 0) {
                bool ml = false, ord = false;
                double tg = 0.0;
                if (act) {
                    const int v = rowvar;
                    if (v < 2 * N) {
                        const int k = v < N ? v : v - N;
                        const double Lk = sl[k], Uk = su[k];
                        const bool fr = Lk == -QINF && Uk == QINF, fx = Lk == Uk;
                        if (v >= N) { if (fr) { ml = true; tg = 0.0; } }
                        else if (fx && !fr) { ml = true; tg = Lk; }
                        ord = !fr && !fx;
                    }
                }
                double best = wave_max_f64(ml ? av : -1.0);
                if (ubool(best > thresh)) {
                    r = wave_first(ml && av == best);
                    leave_val = readlane_f64(tg, r);
                } else {
                    // no equation row can take it: 2x2 principal block pivot through an ordinary pair
                    best = wave_max_f64(ord ? av : -1.0);
                    if (ubool(!(best > thresh))) continue;
                    r = wave_first(ord && av == best);
                    const int v = readlane_i32(rowvar, r);
                    if (v < N) {
                        const double x = readlane_f64(xb, r), plo = udbl(sl[v]), phi = udbl(su[v]);
                        int au;
                        if (x <= plo) { leave_val = plo; au = 0; }
                        else if (x >= phi) { leave_val = phi; au = 1; }
                        else if (plo == -QINF) { leave_val = phi; au = 1; }
                        else if (phi == QINF) { leave_val = plo; au = 0; }
                        else if (phi - x < x - plo) { leave_val = phi; au = 1; }
                        else { leave_val = plo; au = 0; }
                        if (lane == 0) { sat[v] = au; elist[n_enter] = N + v; }
                    } else {
                        leave_val = 0.0;
                        if (lane == 0) elist[n_enter] = v - N;
                    }
                    n_enter++;
                }
            }
            r = uni(r);
            inv = 1.0 / readlane_f64(cm, r);
            delta = (leave_val - readlane_f64(xb, r)) * inv;
        } else {
            const double g = sigma * cm;
            // (LEAN: one Newton step on v_rcp_f64, <= 10 ulp, as in the fused kernels -- the general path keeps the IEEE quotient
            //  of the oracle)
            double rc;
            if constexpr (LEAN) { const double r0 = __builtin_amdgcn_rcp(g); rc = fma(r0, fma(-g, r0, 1.0), r0); }
            else rc = 1.0 / g;
            const bool cndlo = act && g < -ptol && lo > -QINF;
            const bool cndhi = act && g > ptol && hi < QINF;
            const bool cnd = cndlo || cndhi;
            const double arc = cndlo ? -rc : rc;
            const double d = (cndlo ? xb - lo : hi - xb) * arc;
            const double d1 = d + slack * arc;
            double dmax = wave_min_f64(cnd ? d1 : QINF);
            if (self_lim < dmax) dmax = self_lim;
            dmax = udbl(dmax);
            if (ubool(dmax == QINF)) { status = QPN_RAY_TERM; break; }
            const bool cand = cnd && d <= dmax;
            const unsigned long long bal = __ballot(cand);
            if (bal == 0ull) {
                // the entering variable reaches its own far bound first
                const double dl = sigma * self_lim;
                if (act) xb = fma(dl, cm, xb);
                const int ve = (c == XC) ? cNvar : readlane_i32(colvar, c);
                if (ve == 2 * N) {
                    if (c == XC) cNval = 0.0; else if (lane == c) nbval = 0.0;
                    status = QPN_SUCCESS; break;
                }
                const int k = ve;
                const int au = ubool(sigma > 0.0) ? 1 : 0;
                const double nv = udbl(au ? su[k] : sl[k]);
                if (lane == 0) sat[k] = au;
                if (c == XC) cNval = nv; else if (lane == c) nbval = nv;
                pivots++;
                c = col_of(N + k);
                if (c < 0) { status = QPN_FAILURE; break; }
                sigma = au ? -1.0 : 1.0;
                self_lim = QINF;
                if (au) { elo = -QINF; ehi = 0.0; } else { elo = 0.0; ehi = QINF; }   // d_k of an ordinary pair
                wave_sync();
                continue;
            }
            if (__popcll(bal) == 1) {
                r = __ffsll((long long)bal) - 1;
            } else {
                // several rows inside the step bound: largest pivot, the artificial first
                double ag = cand ? fabs(g) : -1.0;
                if (cand && rowvar == 2 * N) ag = QINF;
                const double bestg = wave_max_f64(ag);
                r = wave_first(cand && ag == bestg);
            }
            r = uni(r);
            double step = readlane_f64(d, r);
            if (ubool(step < 0.0)) step = 0.0;
            leave_val = readlane_f64(cndlo ? lo : hi, r);
            inv = sigma * readlane_f64(rc, r);
            delta = sigma * step;
        }

        const int vl = readlane_i32(rowvar, r);
        STAMP(3);   // pivot selection (ratio tests, reductions)
        {   // ---- the pivot: exchange entering column c with the basic variable of row r ----
        const double delta_u = udbl(delta), leave_u = udbl(leave_val), inv_u = udbl(inv);
        const double enter_val = udbl(((c == XC) ? cNval : readlane_f64(nbval, c)) + delta_u);
        const bool rmine = ra == r / BS;
        const bool cmine = (c != XC) && cb == c / BS;
        // raw pivot row -> vrow (scaled by the readers)
        if (rmine) {
            const int rsel = r % BS;
#define M_XR(K, l) if constexpr ((K) < BS && (l) < BS) vrow[cb * PB + (l)] = E(K, l);
#define QPN_LEAF(K) QPN_FOR_L(M_XR, K)
            QPN_DISPATCH8(rsel, QPN_LEAF);
#undef QPN_LEAF
#undef M_XR
        }
        if (lane == r) vrow[NP] = tcol;
        wave_sync();
        {
            // one LDS round trip: extra-column entry of the pivot row, u = pivot column block,
            // v = raw pivot row block
            const double vxr = vrow[NP];
#define M_LDU(k) double u_##k = 0.0; if constexpr ((k) < BS) u_##k = ucol[ra * PB + (k)];
            QPN_FOR_1(M_LDU)
#undef M_LDU
#define M_LDV(l) double v_##l = 0.0; if constexpr ((l) < BS) v_##l = vrow[cb * PB + (l)];
            QPN_FOR_1(M_LDV)
#undef M_LDV
            // row-vector updates (cm is dead afterwards): basic values and the extra column
            {
                const double vx = (c == XC) ? -inv_u : vxr * inv_u;
                double xbn = fma(delta_u, cm, xb);
                double tcn = (c == XC) ? cm * inv_u : fma(-cm, vx, tcol);
                if (lane == r) { xbn = enter_val; tcn = (c == XC) ? inv_u : -vx; }
                xb = xbn; tcol = tcn;
            }
            // the slot of column c carries -inv_u (row fix-up below)
#define M_SCV(l) if constexpr ((l) < BS) { const double sc = v_##l * inv_u; v_##l = (BS * cb + (l) == c) ? -inv_u : sc; }
            QPN_FOR_1(M_SCV)
#undef M_SCV
#define M_FMA(k, l) if constexpr ((k) < BS && (l) < BS) E(k, l) = fma(-u_##k, v_##l, E(k, l));
            QPN_FOR_KL(M_FMA)
#undef M_FMA
            // column c of the new dictionary: T[i][c] = cm_i * inv_u     (in place, owner lanes only)
            if (cmine) {
                const int csel = c % BS;
#define M_FC(k, L)                                                                              \
    if constexpr ((k) < BS && (L) < BS)                                                         \
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(E(k, L)) : "v"(u_##k), "v"(inv_u));
#define QPN_LEAF(L) QPN_FOR_K(M_FC, L)
                QPN_DISPATCH8(csel, QPN_LEAF);
#undef QPN_LEAF
#undef M_FC
            }
            // row r of the new dictionary: T[r][j] = -prow_j, T[r][c] = inv_u (v holds -inv_u there)
            if (rmine) {
                const int rsel = r % BS;
#define M_FR(K, l)                                                                              \
    if constexpr ((K) < BS && (l) < BS)                                                         \
        asm volatile("v_mul_f64 %0, %1, -1.0" : "=v"(E(K, l)) : "v"(v_##l));
#define QPN_LEAF(K) QPN_FOR_L(M_FR, K)
                QPN_DISPATCH8(rsel, QPN_LEAF);
#undef QPN_LEAF
#undef M_FR
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // bookkeeping: row r now holds the entering variable, column c the leaving one
        const int ve = (c == XC) ? cNvar : readlane_i32(colvar, c);
        double nlo = elo, nhi = ehi;          // stage 2: known when the entering variable was chosen
        if (stage_ == 0) interval_uni(ve, N, sl, su, sat, nlo, nhi);
        if (lane == r) { rowvar = ve; lo = nlo; hi = nhi; }
        if (c == XC) { cNvar = vl; cNval = leave_u; }
        else if (lane == c) { colvar = vl; nbval = leave_u; }
        wave_sync();
        }
        STAMP(4);   // rank-1 update
        pivots++;

        if (stage_ == 0) {
            budget--;
            if (n_enter >= 8 * N) stage_ = 1;
        } else {
            if (vl == 2 * N) { status = QPN_SUCCESS; break; }
            int vn;
            if (vl < N) {
                const int k = vl;
                const double Lk = udbl(sl[k]), Uk = udbl(su[k]);
                int au = uni(sat[k]);
                if (ubool(Lk != Uk)) { au = (leave_val == Uk) ? 1 : 0; if (lane == 0) sat[k] = au; }
                vn = N + k;
                sigma = au ? -1.0 : 1.0;
                self_lim = QINF;
                // interval of the entering multiplier d_k (same rule as interval_uni)
                if (ubool(Lk == Uk)) { elo = -QINF; ehi = QINF; }
                else if (ubool(Lk == -QINF && Uk == QINF)) { elo = 0.0; ehi = 0.0; }
                else if (au) { elo = -QINF; ehi = 0.0; }
                else { elo = 0.0; ehi = QINF; }
            } else {
                const int k = vl - N;
                const double Lk = udbl(sl[k]), Uk = udbl(su[k]);
                vn = k;
                sigma = uni(sat[k]) ? -1.0 : 1.0;
                self_lim = Uk - Lk;
                if (ubool(Lk == -QINF && Uk == QINF)) { self_lim = QINF; sigma = 1.0; }
                elo = Lk; ehi = Uk;           // the bounded member p_k enters
            }
            c = col_of(vn);
            if (c < 0) { status = QPN_FAILURE; break; }
            wave_sync();
        }
    }
    STAMP(1);

    // ---- read the point back ----------------------------------------------------------------
    wave_sync();
    gk = (act && a.kind) ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + lane] : 0;
    if (act) { snb[rowvar] = xb; snb[colvar] = nbval; }
    if (lane == 0) snb[cNvar] = cNval;
    wave_sync();
    const double zk = act ? snb[gk ? N + lane : lane] : 0.0;
    wave_sync();
    if (blocked) {
        // reduced problem of a caller that checks on its own original blocks: the point, the status and the pivot count
        if (act) a.z[vo + lane] = zk;
        if (lane == 0) {
            a.status[b] = status;
            if (a.pivots) a.pivots[b] = pivots;
        }
        STAMP(5);
#ifdef QPN_STAMPS
        if (a.stamps && lane == 0)
            for (int i = 0; i < 8; ++i) a.stamps[(size_t)b * 8 + i] = stamp_acc[i];
#endif
        return;
    }
    if (act) ucol[lane] = zk;   // z, broadcast source for the post-check mat-vec (NB <= NP)
    wave_sync();

    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -------------------
    double rk = act ? a.q[vo + lane] : 0.0;
    {
        // 8 independent column loads in flight per step; a zero z_j contributes exactly nothing
        // (M finite), so no per-column branch is needed to match the checker's skip of zeros
        int j = 0;
        for (; j + 8 <= N; j += 8) {
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mv[q8] = act ? Mg[(size_t)(j + q8) * N + lane] : 0.0;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                const double zj = ucol[j + q8];
                rk = (zj != 0.0) ? fma(mv[q8], zj, rk) : rk;
            }
        }
        for (; j < N; ++j) {
            const double zj = ucol[j];
            if (zj != 0.0 && act) rk = fma(Mg[(size_t)j * N + lane], zj, rk);
        }
    }
    const double p = gk ? rk : zk, d = gk ? zk : rk;
    int bad = 0;
    double nres = 0.0;
    unsigned mask = 0;
    if (act) {
        const double lk = sl[lane], uk = su[lane];
        const double tol = a.check_tol;
        if (d > tol && fabs(p - lk) > tol) bad++;
        if (d < -tol && fabs(p - uk) > tol) bad++;
        if (p - lk < -tol) bad++;
        if (p - uk > tol) bad++;
        if (isnan(p) || isnan(d)) bad++;
        double tt = p - d;
        if (tt < lk) tt = lk;
        if (tt > uk) tt = uk;
        nres = fabs(p - tt);
        if (isnan(nres)) nres = QINF;
        const double ct = a.comp_tol;
        auto approx = [&](double x, double y) { return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= ct); };
        const bool eq = approx(lk, uk);
        if (!eq) {
            if (approx(p, lk) && d >= -ct) mask |= 1u;
            if (lk - ct <= p && p <= uk + ct && fabs(d) <= ct) mask |= 2u;
            if (approx(p, uk) && d <= ct) mask |= 4u;
        } else mask = 8u;
        if (gk) mask <<= 4;
    }
    bad = wave_sum_i32(bad);
    nres = wave_max_f64(nres);
    if (bad > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;

    if (act) {
        a.z[vo + lane] = zk;
        if (a.x && lane < a.nd.n) {                                                   // qpn_solve_nodes_into
            const size_t xo = (size_t)b * (size_t)a.stride_x + lane;
            a.x[xo] = zk;
            for (int k = 0; k < a.n_mirror; ++k) a.mirror[k][xo] = zk;
        }
        if (a.active) a.active[vo + lane] = (uint8_t)mask;
    }
    if (lane == 0) {
        a.status[b] = status;
        if (a.resid) a.resid[b] = nres;
        if (a.pivots) a.pivots[b] = pivots;
    }
    STAMP(5);   // read-back + post-check + stores
#ifdef QPN_STAMPS
    if (a.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) a.stamps[(size_t)b * 8 + i] = stamp_acc[i];
#endif
}

// Item selection.  Plain kernel: block b solves item b (optionally gated on only_if[b]).
template <int BS, bool LEAN = false>
__global__ __launch_bounds__(WAVE, 2) void avi_solve_reg(AviBatchArgs a)
{
    const int b = blockIdx.x;
    if (a.only_if && a.only_if[b] != a.only_if_value) return;   // wave-uniform gate
    avi_solve_reg_item<BS, LEAN>(a, b);
}

// Scan kernel (a.scan): the grid is small and fixed; block g looks at items g, g + G, g + 2G, ... (64 at
// a time, one per lane) and solves the ones whose only_if entry matches, one after the other -- the
// fallback pass behind the MFMA kernel costs one near-empty launch instead of `batch` blocks that exit
// at once (and, for qpn_solve_nodes, no separate gated assembly launch).
template <int BS>
__global__ __launch_bounds__(WAVE, 2) void avi_solve_reg_scan(AviBatchArgs a)
{
    const int lane = threadIdx.x;
    const long G = gridDim.x;
    {
        // the common case first: nothing in this block's share is flagged -> leave before any of the
        // solver's set-up code runs
        bool any = false;
        for (long t0 = 0; blockIdx.x + G * t0 < a.batch; t0 += WAVE) {
            const long bb = blockIdx.x + G * (t0 + lane);
            any = any || (bb < a.batch && a.only_if[bb < a.batch ? bb : 0] == a.only_if_value);
        }
        if (__ballot(any) == 0ull) return;
    }
    for (long t0 = 0; blockIdx.x + G * t0 < a.batch; t0 += WAVE) {
        const long bb = blockIdx.x + G * (t0 + lane);
        unsigned long long pick = __ballot(bb < a.batch && a.only_if[bb < a.batch ? bb : 0] == a.only_if_value);
        while (pick) {
            const int i = __ffsll((long long)pick) - 1;
            pick &= pick - 1ull;
            const int b = (int)(blockIdx.x + G * (t0 + i));
            if (a.assemble_first) {
                qpn_assemble_item(a.nd, b, lane, const_cast<double *>(a.M), const_cast<double *>(a.q),
                                  const_cast<double *>(a.l), const_cast<double *>(a.u),
                                  const_cast<uint8_t *>(a.kind));
                __threadfence();
                wave_sync();
            }
            avi_solve_reg_item<BS>(a, b);
            wave_sync();
        }
    }
}

} // namespace

int qpn_avi_reg_block_size(int N)
{
    return N <= 8 ? 1 : N <= 16 ? 2 : N <= 32 ? 4 : N <= 40 ? 5 : N <= 48 ? 6 : N <= 56 ? 7 : 8;
}

hipError_t qpn_launch_avi_solve_reg(const AviBatchArgs &a, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    const dim3 block(WAVE);
    if (a.scan) {
        const dim3 grid((unsigned)(a.batch < 2048 ? a.batch : 2048));
        if (a.N <= 8) hipLaunchKernelGGL(avi_solve_reg_scan<1>, grid, block, 0, stream, a);
        else if (a.N <= 16) hipLaunchKernelGGL(avi_solve_reg_scan<2>, grid, block, 0, stream, a);
        else if (a.N <= 32) hipLaunchKernelGGL(avi_solve_reg_scan<4>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(avi_solve_reg_scan<8>, grid, block, 0, stream, a);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)a.batch);
    if (a.flags & QPN_AVI_IFLAG_BLOCKED_M) {               // reduced problems of the mid-size node path: the lean instantiations
        switch (qpn_avi_reg_block_size(a.N)) {
        case 1: hipLaunchKernelGGL((avi_solve_reg<1, true>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((avi_solve_reg<2, true>), grid, block, 0, stream, a); break;
        case 4: hipLaunchKernelGGL((avi_solve_reg<4, true>), grid, block, 0, stream, a); break;
        case 5: hipLaunchKernelGGL((avi_solve_reg<5, true>), grid, block, 0, stream, a); break;
        case 6: hipLaunchKernelGGL((avi_solve_reg<6, true>), grid, block, 0, stream, a); break;
        case 7: hipLaunchKernelGGL((avi_solve_reg<7, true>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((avi_solve_reg<8, true>), grid, block, 0, stream, a); break;
        }
        return hipGetLastError();
    }
    if (a.N <= 8) hipLaunchKernelGGL(avi_solve_reg<1>, grid, block, 0, stream, a);
    else if (a.N <= 16) hipLaunchKernelGGL(avi_solve_reg<2>, grid, block, 0, stream, a);
    else if (a.N <= 32) hipLaunchKernelGGL(avi_solve_reg<4>, grid, block, 0, stream, a);
    // sizes between the powers of two: fewer dictionary entries per lane (25 / 36 / 49 instead of 64) and a third wave per SIMD
    else if (a.N <= 40) hipLaunchKernelGGL(avi_solve_reg<5>, grid, block, 0, stream, a);
    else if (a.N <= 48) hipLaunchKernelGGL(avi_solve_reg<6>, grid, block, 0, stream, a);
    else if (a.N <= 56) hipLaunchKernelGGL(avi_solve_reg<7>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(avi_solve_reg<8>, grid, block, 0, stream, a);
    return hipGetLastError();
}
