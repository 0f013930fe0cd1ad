// qpn_avi_schur.hip -- MFMA Schur-complement variant of the batched node-AVI solver (gfx950).
//
// For the item shape the hot path is made of -- a node's reduced KKT system
//     kinds = [STD free x n | GAVI x m],  n, m <= 32      (M = [[H, C],[A, D]], q = [g; b])
// -- the crash (Stage A of DESIGN.md section 3: all n free variables enter) is block Gauss-Jordan
// elimination of the H block.  It runs on fp64 matrix cores:
//   * only the TOP half [H | C] (32 x 64, padded) is eliminated, as 2 x 4 tiles of 16 x 16 in the C/D
//     layout of v_mfma_f64_16x16x4_f64 (lane l, reg g  <->  row (l>>4) + 4g, col l&15); an aligned group
//     of 4 rows IS a B operand, so the pivot rows need no data movement at all;
//   * per block pivot (8 of them): the 4 pivot columns go through LDS once, the 4 x 4 pivot block P is
//     LU-factored in registers (uniform, every lane) and each lane solves x L U = u for its own panel row:
//     U' = U P^-1 is the A operand and the RAW pivot rows are the B operand, so the rank-4 update of a tile
//     is ONE instruction (1024 multiply-adds) and the pivot rows become P^-1 V by that same instruction
//     (they carry P - I); no V' = P^-1 V product, which would use 4 of an MFMA's 16 output rows;
//   * the bottom half is never eliminated: with W = H^-1 C and h = H^-1 g in hand,
//         S = D - A W   (32 MFMAs, A operands straight from the staged A block)   and   c = b - A h.
//     This is 25 % fewer flops than eliminating all four quadrants and -- the point -- needs 8 live tiles
//     instead of 16, so FOUR waves fit a SIMD (<= 128 VGPRs, 10 KB LDS each) instead of two.  The solve is a chain of
//     dependent LDS / cross-lane / MFMA latencies, and resident waves are what hides them.
// Stage B (Lemke) then runs on the 32 x 33 dictionary of S in the same tile layout -- a rank-1 exchange per
// pivot, on the VALU (see there) -- and x = -(W lambda + h) is recovered at the end.  Post-check, residual and active-set masks
// are computed on the ORIGINAL blocks exactly as in qpn_avi_reg.hip.
//
// Fused node path (NODES = true, qpn_solve_nodes): M is never materialised.  Qd and Ad are read from HBM
// ONCE with fully coalesced loads into one 8.5 KB LDS buffer (Qd first, to build the H tiles; then Ad,
// which stays for the C tiles, the S = -A W operands and the post-check); the post-check re-reads Qd
// column-wise (lane <-> row: coalesced, served by L2 / Infinity Cache).
//
// Items that do not have this shape, or whose H block fails the no-pivoting test
// (|pivot| >= 1e-4 max(1, max|M|) inside a 4 x 4 block), are flagged (status = -1) and solved by the
// register kernel in ONE compact scan-mode launch -- results identical to the general path.
// Arithmetic differs from the scalar crash only by summation order (block elimination) and one-step
// Newton reciprocals (<= 10 ulp), so primals agree to ~1e-13 and active sets are identical on well-posed items; parity bar:
// DESIGN.md section 2.
#include "qpn_internal.h"
#include <cstdlib>
#include <type_traits>

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
// D = C - A B: on gfx950 the BLGP field of the fp64 MFMAs holds NEG modifiers (bit 0: A, bit 1: B, bit 2: C;
// tools/mfma_neg_probe.hip), so an operand's sign costs no VALU instruction
#define MFMA_NEGA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 1)

// 8 tiles of the top half, named (no arrays: see qpn_avi_reg.hip)
#define TL(I, J) tl_##I##_##J
#define FOR_J(M, I) M(I, 0) M(I, 1) M(I, 2) M(I, 3)
#define FOR_IJ(M) FOR_J(M, 0) FOR_J(M, 1)

struct SchurDebug { double *S, *c, *W, *h; };

// The workgroup IS one wavefront: LDS operations of a wave execute in issue order, so ordering between
// a lane's store and another lane's load needs no s_barrier and no drain of the memory counters -- only
// that the compiler keeps the program order of the LDS accesses (it must: they may alias) and does not
// move them across this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1/x (callers guarantee |x| is well away from 0 on every lane whose result is used).  v_rcp_f64 is good to
// 4.6e-8; one Newton step brings it to <= 2.3e-15 relative (10 ulp), two give the correctly rounded quotient
// (tools/rcp_probe.hip).  The pivoting arithmetic uses one step: its results are certified by the post-check
// on the original blocks, and 1e-15 is far inside the 1e-9 parity bar.
// max(a, |b|) in ONE instruction (fmax(a, fabs(b)) costs three: the compiler canonicalises both operands first; the
// callers feed no NaNs that matter: a NaN entry fails the pivot test and the item goes to the general kernel)
// v[l] + v[l ^ 32] in every lane (gfx950: v_permlane32_swap exchanges the upper half of one register with the lower half of another)
__device__ __forceinline__ double sum_halves(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
__device__ __forceinline__ double min_abs_nc(double a, double b)      // min(a, |b|)
{
    double r;
    asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_abs_nc(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double rcp64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

#define QPN_STG_UNIT 64       /* stagger step of the first round: 64 x 64 clocks = 1.7 us per wave slot */
constexpr int kResidentMI355X = 16 * 256;      // wavefronts of this kernel resident at once: 16 per CU, 256 CUs

// SHAPE: 0 = sizes read at run time (any n, m <= 32); 32 / 16 = every item of the launch is n = m = SHAPE, a compile-time
// constant: no ragged-shape code, bounds predicates or padding selects, and (16) no MFMAs on the padding tiles.
// FULL32 = (SHAPE == 32): the launch's nodes are n = m = 32 -- the hot shape gets its own instantiation, without the
// ragged-shape code, predicates and padding selects (sizes are compile-time constants there)
// SYM: every Qd of the launch is bitwise symmetric (settled once for resident records, qpn_nodes_upload): the lower-left tile
// of H is never loaded or updated -- its pivot columns are the pivot rows of the upper-right tile --, and the lower-left tile
// of S = A H^-1 A' is the transpose of the upper-right one: 8 of the 88 fp64 MFMAs per solve less.
template <bool NODES, int STAGGER = 0, int SHAPE = 0, bool SYM = false>
__global__ __launch_bounds__(WAVE, 4) void avi_solve_schur(AviBatchArgs a, SchurDebug dbg)
{
    static_assert(SHAPE == 0 || SHAPE == 16 || SHAPE == 32, "compile-time shapes: 16 and 32");
    static_assert(!SYM || (NODES && SHAPE == 32), "the symmetric variant exists for n = m = 32 node records");
    static_assert(NODES || SHAPE != 16, "explicit M: N alone settles the split only at N = 64");
    constexpr bool FULL32 = SHAPE == 32;
    constexpr bool HALF16 = SHAPE == 16;
    if constexpr (NODES) {
        // The wavefronts resident from the start of a launch all begin at once: one burst of loads (HBM-bound: the first
        // round's load phase is 3x a later one's), then four waves per SIMD in lock-step through the same phases.
        // Those waves therefore start slot by slot: hardware wave slot s of a SIMD waits s x 1.7 us before its first
        // load (+2.5 % on the 10 000-node launch; later waves are staggered by their predecessors' finishing times).
        // STAGGER = size of that resident set, a template constant (a kernel argument would cost registers in the loops).
        // One asm block (scalar only): no control flow for the register allocator to work around.
        if constexpr (STAGGER > 0) {
            asm volatile("s_cmp_ge_u32 %[bx], %[lim]\n\t"
                         "s_cbranch_scc1 .Lstg_end%=\n\t"
                         "s_getreg_b32 vcc_lo, hwreg(HW_REG_HW_ID, 0, 2)\n"     // wave_id[1:0]: the slot within the SIMD
                         ".Lstg_loop%=:\n\t"
                         "s_cmp_eq_u32 vcc_lo, 0\n\t"
                         "s_cbranch_scc1 .Lstg_end%=\n\t"
                         "s_sleep %[unit]\n\t"                                  // unit x 64 clocks
                         "s_sub_u32 vcc_lo, vcc_lo, 1\n\t"
                         "s_branch .Lstg_loop%=\n"
                         ".Lstg_end%=:"
                         :
                         : [bx] "s"((unsigned)blockIdx.x), [lim] "n"(STAGGER), [unit] "n"(QPN_STG_UNIT)
                         : "scc", "vcc");
        }
    }
    const int N = SHAPE ? 2 * SHAPE : (NODES ? a.nd.n + a.nd.m : a.N);
    const int l = threadIdx.x;
    int b = blockIdx.x;
    if constexpr (NODES) {
        // schedule hint (qpn_order_nodes_by_pivots): this wavefront's node; a bad entry leaves the slot idle
        if (a.order) { b = a.order[blockIdx.x]; if ((unsigned)b >= (unsigned)a.batch) return; }
    }
    const int lc = l & 15, lq = l >> 4;
    // l & 15 recomputed on the spot (one v_and) where the pivot loop compares it: kept in a register for the whole
    // kernel it is what the allocator spills to scratch first
    auto lc_now = [&]() -> int { int t_ = l; asm volatile("" : "+v"(t_)); return t_ & 15; };
    // this wavefront hands its item to the general kernel (status = -1); node path: counted for the owner of the records
    // (the counter's address is read from the kernarg segment on the spot, not kept in registers)
    auto decline = [&]() {
        if (l == 0) {
            a.status[b] = -1;
            if constexpr (NODES) {
                typedef const AviBatchArgs __attribute__((address_space(4))) *kargs_d;
                int32_t *dc = ((kargs_d)__builtin_amdgcn_kernarg_segment_ptr())->decl_count;
                if (dc) atomicAdd(dc, 1);
            }
        }
    };

    // Stage A scratch (sU: pivot columns, [row][k]) and Stage B / read-back scratch never live at the
    // same time: one buffer.
    __shared__ __attribute__((aligned(32))) double sbuf[192];
    double *const sU = sbuf;                    // Stage A: pivot columns of the top half, [32][4]
    double *const sucol = sbuf;                 // Stage B: entering column + extra          [40]
    double *const svrow = sbuf + 40;            // Stage B: pivot row                         [40]
    double *const sval = sbuf;                  // read-back: values by variable id           [66]
    double *const sz = sbuf + 128;              // h, then the solution in item order         [64]
    // fused node path: one block buffer -- Qd while the H tiles are built, then Ad for the rest of the
    // solve, both with column stride 34 (conflict-free tile reads); the two spare slots per column hold
    // q in item order (g and b of the header).  1536 + 8704 = 10240 B: 16 waves per CU.
    constexpr int SQS = 34, SAS = 34;
    __shared__ double sA[32 * SQS];
#define SQ(i) sA[((i) >> 1) * SQS + 32 + ((i) & 1)]

    const double *Mg = NODES ? nullptr : a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)N;
    const bool act = l < N;
#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime() & 0xffffffffull;
#endif
    // node records (fused path): M = [[Qd, -Ad'],[Ad, 0]], q = [qd + R w; B w], src/avi.jl:205-251 + :305-377
    const int nn = SHAPE ? SHAPE : a.nd.n, nm = SHAPE ? SHAPE : a.nd.m, np_ = a.nd.p;
    const double *Q_ = NODES ? a.nd.Qd + (size_t)b * nn * nn : nullptr;
    const double *A_ = NODES ? a.nd.Ad + (size_t)b * nm * nn : nullptr;
    const double *R_ = NODES ? a.nd.R + (size_t)b * nn * np_ : nullptr;
    const double *B_ = NODES ? a.nd.B + (size_t)b * nm * np_ : nullptr;
    const double *w_ = NODES ? a.nd.w + (size_t)b * (size_t)a.nd.stride_w : nullptr;
    auto qelem = [&](int it) -> double {
        if constexpr (NODES) {
            // q = [qd + R w; B w]; the p terms are added in ascending order (same fma chain as the
            // assemble kernel), but loaded eight at a time so that the loads are in flight together
            const bool isx = it < nn;
            const double *col = isx ? R_ + it : B_ + (it - nn);
            const size_t cs = isx ? (size_t)nn : (size_t)nm;
            double s = isx ? a.nd.qd[(size_t)b * nn + it] : 0.0;
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
            return s;
        } else {
            return a.q[vo + it];
        }
    };
    // bounds / kind of item row l (re-read where needed rather than kept in registers)
    auto row_bounds = [&](double &lk_, double &uk_, int &gk_) {
        lk_ = 0.0; uk_ = 0.0; gk_ = 0;
        if constexpr (NODES) {
            if (act) {
                if (l < nn) { lk_ = -QINF; uk_ = QINF; }
                else { lk_ = a.nd.l[(size_t)b * nm + (l - nn)]; uk_ = a.nd.u[(size_t)b * nm + (l - nn)]; gk_ = 1; }
            }
        } else {
            lk_ = act ? a.l[vo + l] : 0.0; uk_ = act ? a.u[vo + l] : 0.0;
            gk_ = (act && a.kind) ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + l] : 0;
        }
    };

    // ---- structure test: leading free STD rows, then GAVI rows -------------------------------------
    int n_s, m_s;
    if constexpr (NODES) {
        // a node record IS of that shape (n free rows, then m GAVI rows): nothing to read, so the block loads
        // below are the first memory round trip of the wave
        n_s = nn; m_s = nm;
        if (!(n_s + m_s == N && n_s <= 32 && m_s <= 32 && n_s >= 1)) { decline(); return; }
    } else {
        double lk, uk; int gk;
        row_bounds(lk, uk, gk);
        const bool isfree = act && !gk && lk == -QINF && uk == QINF;
        const unsigned long long mfree = qpn_ballot(isfree), mg = qpn_ballot(act && gk);
        n_s = __popcll(mfree); m_s = __popcll(mg);
        // (FULL32: the launch's items are N = 64, and n + m = 64 with n, m <= 32 leaves n = m = 32: constants from here on)
        const bool shape_ok = n_s + m_s == N && n_s <= 32 && m_s <= 32 && n_s >= 1 && (!FULL32 || a.N == 64) &&
                              mfree == ((n_s >= 64) ? ~0ull : ((1ull << n_s) - 1ull)) &&
                              mg == ((((m_s + n_s) >= 64) ? ~0ull : ((1ull << (m_s + n_s)) - 1ull)) & ~((1ull << n_s) - 1ull));
        if (!shape_ok) { decline(); return; }
    }

    const int n = SHAPE ? SHAPE : n_s, m = SHAPE ? SHAPE : m_s;     // (compile-time constants in the fixed-shape instantiations)

    // ---- load: the top half [H | C] straight into the MFMA tile layout ----------------------------------
#define M_DECL(I, J) d4 TL(I, J);
    FOR_IJ(M_DECL)
#undef M_DECL
    double mabs = 0.0;
    if constexpr (NODES) {
        // Qd and Ad with fully coalesced loads (two columns of 32 rows per instruction = 512 contiguous
        // bytes when n = m = 32), all 32 + the q loads in flight at once.  FULL (n = m = 32, the hot shape)
        // drops every bounds predicate, clamp and padding select.
        const int r50 = l & 31, ch0 = l >> 5;
#define M_LOADH(I, J, FULL)                                                                         \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq, cc = 16 * (J) + lc;                                   \
        double v = sA[cc * SQS + rr];                                                               \
        if (!(FULL) && rr == cc && rr >= n) v = 1.0;                /* padded x rows: identity */     \
        TL(I, J)[g] = v;                                                                            \
    }
#define M_LOADC(I, J, FULL)                                                                         \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq, ck = 16 * ((J) - 2) + lc;                             \
        const double t_ = sA[rr * SAS + ck];        /* the tiles hold -C = +Ad': W~ = -W, signs folded in below */ \
        TL(I, J)[g] = ((FULL) || (rr < n && ck < m)) ? t_ : 0.0;                                    \
    }
#define M_NODE_LOAD(FULL)                                                                           \
    {                                                                                               \
        double vq[16], va[16];                                                                      \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                            \
            const int cj = 2 * t + ch0;                                                             \
            const bool okq = (FULL) || (cj < nn && r50 < nn), oka = (FULL) || (cj < nn && r50 < nm); \
            vq[t] = Q_[okq ? (size_t)cj * ((FULL) ? 32 : nn) + r50 : 0];                            \
            va[t] = A_[oka ? (size_t)cj * ((FULL) ? 32 : nm) + r50 : 0];                            \
        }                                                                                           \
        SQ(l) = act ? qelem(l) : 0.0;                                                               \
        /* (ragged shapes: the bounds predicates are formed again from a fresh copy of the lane id where the values are    \
           staged -- carried over from the loads they are 32 lane masks held across the round trip) */ \
        int l2_ = l; asm volatile("" : "+v"(l2_));                                                  \
        const int r5 = l2_ & 31, ch = l2_ >> 5;                                                     \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                            \
            const int cj = 2 * t + ch;                                                              \
            const double q_ = ((FULL) || (cj < nn && r5 < nn)) ? vq[t] : 0.0;                       \
            sA[cj * SQS + r5] = q_;                                                                 \
            mabs = max_abs_nc(mabs, q_);                                                            \
        }                                                                                           \
        wave_sync();                                                                                \
        M_LOADH(0, 0, FULL) M_LOADH(0, 1, FULL) if constexpr (!SYM) M_LOADH(1, 0, FULL)             \
        M_LOADH(1, 1, FULL)                                                                         \
        wave_sync();                                                                                \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                            \
            const int cj = 2 * t + ch;                                                              \
            const double a_ = ((FULL) || (cj < nn && r5 < nm)) ? va[t] : 0.0;                       \
            sA[cj * SAS + r5] = a_;                                                                 \
            mabs = max_abs_nc(mabs, a_);                                                            \
        }                                                                                           \
        wave_sync();                                                                                \
        M_LOADC(0, 2, FULL) M_LOADC(0, 3, FULL) M_LOADC(1, 2, FULL) M_LOADC(1, 3, FULL)             \
    }
        if constexpr (FULL32) M_NODE_LOAD(true) else M_NODE_LOAD(false)      // (n = m = 32 launches take the FULL32 instantiation)
#undef M_NODE_LOAD
#undef M_LOADH
#undef M_LOADC
    } else {
        // Explicit M (N x N column-major, item order = [x | lambda]): the four blocks H = M[0:n, 0:n], C = M[0:n, n:N],
        // A = M[n:N, 0:n] go through the same LDS block buffer as the node path's Qd and Ad -- whole columns of 32 rows with
        // coalesced loads (256 contiguous bytes per column, two columns per instruction), every byte read once; A stays in the
        // buffer (the A operands of the S product and c = b - A h).  D = M[n:N, n:N] is read where it is used (straight into
        // the S accumulators): its share of max |M| is settled there, see `minpiv`.
        // (ragged shapes: every batch of bounds predicates is formed from a fresh copy of the lane id -- carried over from the
        // loads to the staging they are 48 lane masks held across the round trips)
        auto lane_now = [&]() -> int { int t_ = l; asm volatile("" : "+v"(t_)); return t_; };
        double vh[16], vc[16];
        {
        const int l0_ = lane_now(), r5 = l0_ & 31, ch = l0_ >> 5;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int cj = 2 * t + ch;
            vh[t] = Mg[(cj < n && r5 < n) ? (size_t)cj * N + r5 : 0];
            vc[t] = Mg[(cj < m && r5 < n) ? (size_t)(n + cj) * N + r5 : 0];
        }
        }
        {
        const int l0_ = lane_now(), r5 = l0_ & 31, ch = l0_ >> 5;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int cj = 2 * t + ch;
            const double h_ = (cj < n && r5 < n) ? vh[t] : 0.0;
            sA[cj * SQS + r5] = h_;
            mabs = max_abs_nc(mabs, h_);
        }
        }
        wave_sync();
#define M_LOADH(I, J)                                                                               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq, cc = 16 * (J) + lc;                                   \
        double v = sA[cc * SQS + rr];                                                               \
        if (rr == cc && rr >= n) v = 1.0;                           /* padded x rows: identity */     \
        TL(I, J)[g] = v;                                                                            \
    }
        M_LOADH(0, 0) M_LOADH(0, 1) M_LOADH(1, 0) M_LOADH(1, 1)
#undef M_LOADH
        wave_sync();
        {
        const int l0_ = lane_now(), r5 = l0_ & 31, ch = l0_ >> 5;
#pragma unroll
        for (int t = 0; t < 16; ++t) {          // A: requested now (into H's registers), staged after C
            const int cj = 2 * t + ch;
            vh[t] = Mg[(cj < n && r5 < m) ? (size_t)cj * N + n + r5 : 0];
        }
        }
        {
        const int l0_ = lane_now(), r5 = l0_ & 31, ch = l0_ >> 5;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int cj = 2 * t + ch;
            const double c_ = (cj < m && r5 < n) ? vc[t] : 0.0;
            sA[cj * SQS + r5] = c_;
            mabs = max_abs_nc(mabs, c_);
        }
        }
        wave_sync();
#define M_LOADC(I, J)                                                                               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq, ck = 16 * ((J) - 2) + lc;                             \
        TL(I, J)[g] = sA[ck * SQS + rr];                            /* zeros outside n x m already */ \
    }
        M_LOADC(0, 2) M_LOADC(0, 3) M_LOADC(1, 2) M_LOADC(1, 3)
#undef M_LOADC
        wave_sync();
        {
        const int l0_ = lane_now(), r5 = l0_ & 31, ch = l0_ >> 5;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int cj = 2 * t + ch;
            const double a_ = (cj < n && r5 < m) ? vh[t] : 0.0;
            sA[cj * SAS + r5] = a_;                                  // [column of x][constraint row]: the node path's Ad layout
            mabs = max_abs_nc(mabs, a_);
        }
        }
        // (the first reader of A is behind the wave_syncs of Stage A)
    }
    // extra column: g = q of the x rows, lane l <-> row l of the top half (lanes >= 32 idle)
    double kx;
    if constexpr (NODES) kx = (l < n) ? SQ(l) : 0.0; else kx = (l < n) ? qelem(l) : 0.0;
    double *const sP = sbuf + 128;              // Stage A: the 4 x 4 pivot block of the current step, raw (the sz block is idle)
    const double mscale = wave_max_f64(mabs);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);

    STAMP(0);   // setup + load
    // ---- Stage A: 8 rank-4 block pivots of the top half on the matrix cores ------------------------------
    // Pivot rows carry P - I in sU so that the update T -= (U P^-1) V turns them into P^-1 V themselves:
    // no separate V' = P^-1 V product (an MFMA that would use 4 of its 16 output rows) is needed.
    bool fail = false;
    // explicit M: the pivot threshold is relative to max |M| over the WHOLE item, and the D block is read only after the
    // crash.  The smallest accepted pivot is kept; once max |D| is known, an item with a pivot below 1e-4 max |D| is declined
    // after all -- the same items as a threshold known in advance would have declined (the pivots do not depend on it).
    // (kept in an idle LDS slot, not in a register pair across the eight steps)
    double *const sMinPiv = sbuf + 191;
    if constexpr (!NODES) { if (l == 0) sMinPiv[0] = QINF; }
    // NEGF (node path): the factorisation carries the opposite sign throughout -- nrd = -1 / u_ss, the multipliers kept in pm are
    // -l_is, the panel solve yields -U' -- so that every step is a plain fma and no operand is negated by an instruction of
    // its own (the compiler materialised three negated copies per step); bit for bit the same numbers.  The explicit-M
    // instantiations keep the plain signs (there the sign-flipped form costs registers: the ragged one spills).
    constexpr bool NEGF = NODES;
#define FMS(a_, b_, c_) (NEGF ? fma((a_), (b_), (c_)) : fma(-(a_), (b_), (c_)))
#define M_GATHER(I, JP, GP)                                                                         \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq;                                                       \
        double v = TL(I, JP)[g];                                                                    \
        /* row p0 + kcol sits in tile JP, register GP, lane group lq == kcol: a static test */      \
        if ((I) == (JP) && g == (GP)) sP[lq * 4 + kcol] = v;        /* the pivot block itself, raw (16 lanes) */ \
        if ((I) == (JP) && g == (GP) && lq == kcol) v -= 1.0;                                       \
        sU[rr * 4 + kcol] = v;                                                                      \
    }
#define M_COLTILE(J, IP, GP)                                                                        \
    {                                                                                               \
        /* the 4 pivot rows of this column tile are a B operand; a private copy, because they are also part \
           of the accumulator tile TL(IP, J) (else the compiler moves the whole tile out and back) */ \
        double vraw = TL(IP, J)[GP];                                                                \
        asm volatile("" : "+v"(vraw));      /* opaque: the copy itself is the compiler's (hazard-aware) */ \
        TL(0, J) = NEGF ? MFMA(au0, vraw, TL(0, J)) : MFMA_NEGA(au0, vraw, TL(0, J));   /* T -= U' V */ \
        if constexpr (!HALF16 && !(SYM && (J) == 0))                                                \
            TL(1, J) = NEGF ? MFMA(au1, vraw, TL(1, J)) : MFMA_NEGA(au1, vraw, TL(1, J));     /* (16: rows 16.. are padding) */ \
    }
#define M_USOLVE                                                                                    \
        if (l < 32) {                                                                           \
            const int lr_ = l;                                                                  \
            const d4 ur = *reinterpret_cast<const d4 *>(sU + lr_ * 4);                          \
            /* (NEGF: n_k = -y_k, up = -U') */                                                  \
            const double n0 = ur[0] * nrd[0];                                                   \
            const double n1 = FMS(n0, pm[0][1], ur[1]) * nrd[1];                                \
            const double n2 = FMS(n1, pm[1][2], FMS(n0, pm[0][2], ur[2])) * nrd[2];             \
            const double n3 = FMS(n2, pm[2][3], FMS(n1, pm[1][3], FMS(n0, pm[0][3], ur[3]))) * nrd[3]; \
            d4 up;                                                                              \
            up[3] = n3;                                                                         \
            up[2] = FMS(up[3], pm[3][2], n2);                                                   \
            up[1] = FMS(up[3], pm[3][1], FMS(up[2], pm[2][1], n1));                             \
            up[0] = FMS(up[3], pm[3][0], FMS(up[2], pm[2][0], FMS(up[1], pm[1][0], n0)));       \
            {                                                                                   \
                const double t_ = fma(up[3], x3, fma(up[2], x2, fma(up[1], x1, up[0] * x0)));   \
                kx = NEGF ? kx + t_ : kx - t_;                                                  \
            }                                                                                   \
            *reinterpret_cast<d4 *>(sU + lr_ * 4) = up;                                         \
        }
#define M_STEP(KB, JP, GP)                                                                          \
    if (!fail && 4 * (KB) < n) {       /* a block of padded rows is an identity pivot: nothing moves */ \
        constexpr int p0 = 4 * (KB);                                                                \
        constexpr int cq = p0 & 15;                                                                 \
        const int kcol = lc - cq;                                                                   \
        if constexpr (SYM && (JP) == 0) {                                                           \
            /* rows 16..31 of the pivot columns = the pivot rows' entries in tile (0, 1): one store per lane */ \
            if (kcol >= 0 && kcol < 4) { M_GATHER(0, JP, GP) }                                      \
            sU[(16 + lc) * 4 + lq] = TL(0, 1)[GP];                                                  \
        } else {                                                                                    \
            if (kcol >= 0 && kcol < 4) { M_GATHER(0, JP, GP) M_GATHER(1, JP, GP) }                  \
        }                                                                                           \
        wave_sync();                                                                                \
        /* P = L U (unit lower L, no pivoting; uniform, every lane) */                              \
        double pm[4][4];                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
            const d4 row = *reinterpret_cast<const d4 *>(sP + i * 4);                               \
            pm[i][0] = row[0]; pm[i][1] = row[1]; pm[i][2] = row[2]; pm[i][3] = row[3];             \
        }                                                                                           \
        /* (tried twice: these four entries of the extra column through LDS with the pivot block instead of eight v_readlane: \
           0.8 - 0.9 % slower) */                                                                   \
        const double x0 = readlane_f64(kx, p0), x1 = readlane_f64(kx, p0 + 1);                      \
        const double x2 = readlane_f64(kx, p0 + 2), x3 = readlane_f64(kx, p0 + 3);                  \
        bool okp = true;                                                                            \
        double minpiv = QINF;                                                                       \
        double nrd[4];                                  /* 1 / u_ss (NEGF: -1 / u_ss) */            \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                             \
            okp = okp && fabs(pm[s][s]) >= diag_thr;                                                \
            if constexpr (!NODES) minpiv = min_abs_nc(minpiv, pm[s][s]);                            \
            nrd[s] = NEGF ? rcp64(-pm[s][s]) : rcp64(pm[s][s]);                                     \
            _Pragma("unroll") for (int i = s + 1; i < 4; ++i) {                                     \
                const double f = pm[i][s] * nrd[s];     /* l_is (NEGF: -l_is), kept in place */     \
                pm[i][s] = f;                                                                       \
                _Pragma("unroll") for (int j = s + 1; j < 4; ++j) pm[i][j] = FMS(f, pm[s][j], pm[i][j]); \
            }                                                                                       \
        }                                                                                           \
        /* U' = U P^-1, row l of the panel: solve x L U = u (pivot rows hold P - I, so theirs   \
           is I - P^-1), and the extra column kx_l -= U'[l] . x_piv */                          \
        /* (tried: the solve on every lane and ahead of the branch on the pivot test, so that it interleaves with the \
           factorisation -- 0.3 % slower) */                                                        \
        if (!ubool(okp)) { fail = true; }                                                           \
        else {                                                                                      \
            if constexpr (!NODES) { if (l == 0) sMinPiv[0] = fmin(sMinPiv[0], minpiv); }            \
            M_USOLVE                                                                                \
            wave_sync();                                                                            \
            const double au0 = sU[(0 + lc) * 4 + lq], au1 = sU[(16 + lc) * 4 + lq];    /* U' (NEGF: -U') */    \
            if ((JP) <= 0) M_COLTILE(0, JP, GP)                                                     \
            if constexpr (!HALF16) { if ((JP) <= 1) M_COLTILE(1, JP, GP) }     /* (16: columns 16.. of H and of C are zero) */ \
            M_COLTILE(2, JP, GP)                                                                    \
            if constexpr (!HALF16) M_COLTILE(3, JP, GP)                                             \
            wave_sync();                                                                            \
        }                                                                                           \
    }
    M_STEP(0, 0, 0) M_STEP(1, 0, 1) M_STEP(2, 0, 2) M_STEP(3, 0, 3)
    M_STEP(4, 1, 0) M_STEP(5, 1, 1) M_STEP(6, 1, 2) M_STEP(7, 1, 3)
#undef M_STEP
#undef M_USOLVE
#undef FMS
#undef M_COLTILE
#undef M_GATHER
    if (fail) { decline(); return; }

    // ---- S = D - A W on the matrix cores, c = b - A h -----------------------------------------------------
    // W = TL(0..1, 2..3) (rows = x, an aligned group of 4 rows is a B operand), h = kx (lanes 0..31).
#define SB(Ib, Jb) sb_##Ib##_##Jb
    d4 SB(0, 0), SB(0, 1), SB(1, 0), SB(1, 1);
    if (l < 32) sz[l] = kx;
    if constexpr (NODES) {
        const d4 z4 = {0.0, 0.0, 0.0, 0.0};
        SB(0, 0) = z4; SB(0, 1) = z4; SB(1, 0) = z4; SB(1, 1) = z4;
    } else {
#define M_LOADD(Ib, Jb)                                                                             \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rk_ = 16 * (Ib) + 4 * g + lq, ck = 16 * (Jb) + lc;                                \
        const bool valid = rk_ < m && ck < m;                                                       \
        const double v = Mg[valid ? (size_t)(n + ck) * N + n + rk_ : 0];                            \
        SB(Ib, Jb)[g] = valid ? v : 0.0;                                                            \
    }
        M_LOADD(0, 0) M_LOADD(0, 1) M_LOADD(1, 0) M_LOADD(1, 1)
#undef M_LOADD
        double md = 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            md = max_abs_nc(md, SB(0, 0)[g]); md = max_abs_nc(md, SB(0, 1)[g]);
            md = max_abs_nc(md, SB(1, 0)[g]); md = max_abs_nc(md, SB(1, 1)[g]);
        }
        if (ubool(sMinPiv[0] < 1e-4 * wave_max_f64(md))) { decline(); return; }
    }
    // A operand (16 x 4) of row tile Ib, k-block kk: element (i = lc, k = lq) = A[16 Ib + lc][4 kk + lq].
    // Node path: the tiles hold W~ = -W, so S = D - A W = D + A W~ (plain MFMA); M path: S = D - A W (NEG on A).
    auto aop = [&](int Ib, int kk) -> double {
        const int rk_ = 16 * Ib + lc, cj = 4 * kk + lq;
        return sA[cj * SAS + rk_];                                  // zeros outside m x n already
    };
#define M_SACC(a_, b_, c_) (NODES ? MFMA(a_, b_, c_) : MFMA_NEGA(a_, b_, c_))
#define M_SK(I, g, kk)                                                                              \
    if (4 * (kk) < n) {                /* rows of W beyond n are zero */                             \
        const double a0_ = aop(0, kk), a1_ = aop(1, kk);                                            \
        SB(0, 0) = M_SACC(a0_, TL(I, 2)[g], SB(0, 0));                                              \
        if constexpr (!HALF16) {           /* (16: the other three tiles of S are padding, zero) */    \
            SB(0, 1) = M_SACC(a0_, TL(I, 3)[g], SB(0, 1));                                          \
            if constexpr (!SYM) SB(1, 0) = M_SACC(a1_, TL(I, 2)[g], SB(1, 0));                      \
            SB(1, 1) = M_SACC(a1_, TL(I, 3)[g], SB(1, 1));                                          \
        }                                                                                           \
    }
    // bounds of pair l for Stage B: requested here so that the round trip hides behind the 32 MFMAs
    double lo_pre = -QINF, hi_pre = QINF;
    if constexpr (NODES) {
        if (l < m) { lo_pre = a.nd.l[(size_t)b * nm + l]; hi_pre = a.nd.u[(size_t)b * nm + l]; }
    }
    M_SK(0, 0, 0) M_SK(0, 1, 1) M_SK(0, 2, 2) M_SK(0, 3, 3) M_SK(1, 0, 4) M_SK(1, 1, 5) M_SK(1, 2, 6) M_SK(1, 3, 7)
#undef M_SK
#undef M_SACC
    if constexpr (SYM) {
        // S(1,0) = S(0,1)'.  (Also exact and 0.5 % slower: on the matrix cores -- register kb of a tile in the accumulator layout IS
        // the A operand of its transpose's k-block kb, element (i = lc, k = lq) = X[4 kb + lq][lc], with B = rows 4 kb .. 4 kb + 3
        // of the identity: X' = sum_kb MFMA(X[kb], I[4 kb .. 4 kb + 3][.]).)
        // Here: through the idle Stage A block of sbuf (128 doubles), eight rows per round; row r of a round sits at
        // r * 16 + ((c + 4 (r >> 1)) & 15): the writes fill four whole rows, the 32 reading lanes (lc >> 3 == round) hit 32 banks
        const int rr_ = lc & 7, rot_ = 4 * (rr_ >> 1);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int r = 4 * gg + lq;
                sbuf[r * 16 + ((lc + 4 * (r >> 1)) & 15)] = SB(0, 1)[2 * hh + gg];
            }
            wave_sync();
            if ((lc >> 3) == hh) {
#pragma unroll
                for (int g2 = 0; g2 < 4; ++g2) SB(1, 0)[g2] = sbuf[rr_ * 16 + ((4 * g2 + lq + rot_) & 15)];
            }
            wave_sync();
        }
    }
    wave_sync();
    // c_k = b_k - sum_j A[k][j] h_j, lane k <-> pair k (k < m)
    double xb = 0.0;
    {
        const bool lowr = l < m;
        {
            // both halves of the wavefront work on the (up to) 32 rows: lane l and lane l + 32 take half of the columns each of row
            // l & 31 (columns beyond n are zero in the buffer; n = m = 16: 16 columns in all)
            constexpr int HC = HALF16 ? 8 : 16;
            const int ls = l & 31, j0 = (l >> 5) * HC;
            double acc = 0.0;
            if constexpr (NODES) acc = lowr ? SQ(n + ls) : 0.0; else acc = lowr ? qelem(n + ls) : 0.0;
            double acc2 = 0.0;
#pragma unroll 8
            for (int j = 0; j < HC; j += 2) {
                acc = fma(-sA[(j0 + j) * SAS + ls], sz[j0 + j], acc);
                acc2 = fma(-sA[(j0 + j + 1) * SAS + ls], sz[j0 + j + 1], acc2);
            }
            const double both = sum_halves(acc + acc2);        // (every lane takes part: not inside the select)
            xb = lowr ? both : 0.0;
        }
    }
    STAMP(6);   // crash on the matrix cores

    if (dbg.S) {
        // diagnostic builds: dump S (32x32), c, W (32x32), h in row-major
#define M_DUMPW(I, J)                                                                               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (I) + 4 * g + lq, cc = 16 * ((J) - 2) + lc;                             \
        dbg.W[(size_t)b * 1024 + (size_t)rr * 32 + cc] = TL(I, J)[g];                               \
    }
#define M_DUMPS(Ib, Jb)                                                                             \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
        const int rr = 16 * (Ib) + 4 * g + lq, cc = 16 * (Jb) + lc;                                 \
        dbg.S[(size_t)b * 1024 + (size_t)rr * 32 + cc] = SB(Ib, Jb)[g];                             \
    }
        M_DUMPW(0, 2) M_DUMPW(0, 3) M_DUMPW(1, 2) M_DUMPW(1, 3)
        M_DUMPS(0, 0) M_DUMPS(0, 1) M_DUMPS(1, 0) M_DUMPS(1, 1)
#undef M_DUMPW
#undef M_DUMPS
        if (l < 32) { dbg.h[(size_t)b * 32 + l] = kx; dbg.c[(size_t)b * 32 + l] = xb; }
        if (l == 0) a.status[b] = -2;
        return;
    }

    // ================= Stage B: Lemke on the Schur dictionary (32 pairs, tile layout) =================
    // pair k (k < 32) <-> item row n + k:  p_k = s_k = (S lambda + c)_k in [l_k, u_k],  d_k = lambda_k.
    // ids: p_k -> k, d_k -> 32 + k, artificial -> 64; column index 32 = the extra (covering) column.
    // The loop is bound by instruction issue.  On gfx950 an fp64 MFMA occupies the SIMD's fp64 pipe for as
    // long as the 16 v_fma_f64 it replaces and does not overlap with other VALU work
    // (tools/mfma_valu_overlap.hip), so a RANK-1 exchange -- which would waste 3 of the 4 K slots -- is done
    // with v_fma_f64: 16 per lane, in place, plus lane-masked fix-ups of the pivot column (8 v_mul_f64 on
    // the 4 owner lanes) and the pivot row (2 v_mov_b64 on the 16 owner lanes).  The whole exchange is ONE
    // asm block: the compiler sees 16 read-modify-write registers, so the dictionary is never copied,
    // selected or spilled whatever the control flow around it; EXEC narrowing and the uniform
    // column-tile / row-register dispatch live inside the block.
    // Pair bounds / bound flags sit in lane k and are fetched with v_readlane; the only LDS traffic per
    // pivot is the entering column and the pivot row.
    constexpr int NBP = 32, XC = 32, VTH = 64;
    const bool actb = l < NBP;
    double lo = -QINF, hi = QINF;
    {
        const int it = l < m ? n + l : -1;
        if constexpr (NODES) { lo = lo_pre; hi = hi_pre; }
        else if (actb && it >= 0) { lo = a.l[vo + it]; hi = a.u[vo + it]; }
    }
    // equality GAVI rows need their multiplier crashed in: left to the general kernel
    if (qpn_ballot(actb && lo == hi)) { decline(); return; }
    const double lo0 = lo, hi0 = hi;      // bounds of pair l (fixed); lo/hi follow the row's basic variable
    // class of pair l (0 bounded on at least one side, 2 free; equal bounds were sent to the general kernel
    // above) and its range, so that the per-pivot bookkeeping is integer / scalar work, not fp64 compares
    const int clsv = (lo0 == -QINF && hi0 == QINF) ? 2 : 0;
    const double rngv = hi0 - lo0;
    int satv = 0;                         // pair l: 1 = bounded variable rests at its upper bound
    // row vectors live in lanes 0..31; column vectors (colvar, nbval) in lanes 0..32: lane 32 is the extra
    // (covering) column, so no column is a special case in the bookkeeping
    int rowvar = actb ? l : -1, colvar = actb ? NBP + l : (l == XC ? VTH : -1);
    double nbval = 0.0, tcol = 0.0;
    wave_sync();                      // Stage A and the c sweep are done with sbuf

    // the dictionary as two 8-vectors per lane: element 4 Ib + g of SJ<Jb> = row 16 Ib + 4 g + lq, column
    // 16 Jb + lc.  Static element accesses are plain registers; the pivot row is read with a wave-uniform
    // DYNAMIC index (s_set_gpr_idx_on + v_mov: no branch tree, no scratch).
    typedef double d8 __attribute__((ext_vector_type(8)));
    d8 SJ0 = {SB(0, 0)[0], SB(0, 0)[1], SB(0, 0)[2], SB(0, 0)[3], SB(1, 0)[0], SB(1, 0)[1], SB(1, 0)[2], SB(1, 0)[3]};
    d8 SJ1 = {SB(0, 1)[0], SB(0, 1)[1], SB(0, 1)[2], SB(0, 1)[3], SB(1, 1)[0], SB(1, 1)[1], SB(1, 1)[2], SB(1, 1)[3]};
#define SD(Ib, Jb, g) SJ##Jb[4 * (Ib) + (g)]
    auto col_of = [&](int v) -> int { return wave_first(colvar == v); };     // ids are >= 0, idle lanes hold -1
    int pivots = n;                       // the crash brought n free variables in (Stage A)
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;
    int status = QPN_FAILURE;
    int c = XC;
    bool sneg = true;                     // direction of the entering variable: sigma = sneg ? -1 : +1
    double self_lim = 0.0, elo = 0.0, ehi = QINF;
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
            if (l == XC) nbval = theta0;
            self_lim = theta0;
            status = QPN_MAX_ITERS;
        }
    }
    while (status == QPN_MAX_ITERS) {
        if (pivots >= max_piv) break;
        c = uni(c);
        // ---- entering column -> sucol, permuted so that a lane group reads its 8 rows as 8 consecutive doubles
        // (row r = q + 4 g + 16 Ib sits at q*8 + 4 Ib + g)
        if (c == XC) { if (actb) sucol[(l & 3) * 8 + ((l >> 4) << 2) + ((l >> 2) & 3)] = tcol; }
        else if (lc_now() == (c & 15)) {
            d4 *const dst = reinterpret_cast<d4 *>(sucol + lq * 8);
            if (c < 16) { dst[0] = d4{SD(0, 0, 0), SD(0, 0, 1), SD(0, 0, 2), SD(0, 0, 3)}; dst[1] = d4{SD(1, 0, 0), SD(1, 0, 1), SD(1, 0, 2), SD(1, 0, 3)}; }
            else { dst[0] = d4{SD(0, 1, 0), SD(0, 1, 1), SD(0, 1, 2), SD(0, 1, 3)}; dst[1] = d4{SD(1, 1, 0), SD(1, 1, 1), SD(1, 1, 2), SD(1, 1, 3)}; }
        }
        wave_sync();
        const double cm = actb ? sucol[(l & 3) * 8 + ((l >> 4) << 2) + ((l >> 2) & 3)] : 0.0;
        STAMP(1);   // entering column through LDS
        // ---- ratio test (same rule as the general kernel; reciprocal by Newton instead of a division)
        const double gdir = __hiloint2double(__double2hiint(cm) ^ (sneg ? (int)0x80000000 : 0), __double2loint(cm));
        const double rc = rcp64(gdir);
        // the bound the row's basic variable moves towards: lo when it decreases (gdir < 0), hi when it increases; a row is a
        // candidate when that bound is finite and |gdir| is above the pivot tolerance.  (t - xb) * rc = |t - xb| * |rc|, both
        // factors negated exactly: the same bits as the two-sided form, with one subtraction and two compares less
        const bool gneg = __double2hiint(gdir) < 0;
        const double tb = gneg ? lo : hi;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wbitwise-instead-of-logical"
        const bool cnd = (actb & (fabs(gdir) > ptol)) & (fabs(tb) < QINF);       // (no short circuit: three lane masks and two s_and)
#pragma clang diagnostic pop
        const double arc = fabs(rc);
        // the ratio of a row that is no candidate is +inf from here on (inf stays inf through the slack term), so neither the
        // minimum nor the tie set needs the candidate flag again: the tie set is ONE compare
        const double dd = cnd ? (tb - xb) * rc : QINF;
        const double d1 = fma(slack, arc, dd);
        const double dmax = wave_min32_with_limit_f64(d1, self_lim);         // wave-uniform (an SGPR pair)
        if (uni(__double2hiint(dmax)) == 0x7ff00000) { status = QPN_RAY_TERM; break; }      // +inf: nothing blocks the ray
        const unsigned long long bal = qpn_ballot(dd <= dmax);
        // Both outcomes below end in the SAME exchange block (a bound flip runs it with v = 0 and empty lane
        // masks: a no-op), so the dictionary registers have one definition per iteration.
        double v0 = 0.0, v1 = 0.0, inv = 0.0;
        unsigned long long mcol = 0ull, mrow = 0ull;
        int cx = XC, rsel = 0, cnext = -1;
        // ... and the same single-lane writes of the bookkeeping vectors (a write that does not apply goes to
        // idle lane 63), so those registers are never copied around a branch either
        int rW = 63, veW = 0, cW = 63, vlW = 0, kW = 63, auW = 0;
        double eloW = 0.0, ehiW = 0.0, nbW = 0.0;
        bool stop = false;
        if (bal == 0ull) {
            // the entering variable reaches its own opposite bound first: no basis change
            const double dl = sneg ? -self_lim : self_lim;
            if (actb) xb = fma(dl, cm, xb);
            const int ve = readlane_i32(colvar, c);
            if (ve == VTH) { nbW = 0.0; status = QPN_SUCCESS; stop = true; }
            else {
                const int k = ve;
                const int au = sneg ? 0 : 1;
                nbW = au ? readlane_f64(hi0, k) : readlane_f64(lo0, k);
                kW = k; auW = au;
                pivots++;
                cnext = col_of(NBP + k);
                if (cnext < 0) { status = QPN_FAILURE; stop = true; }
                sneg = au != 0;
                self_lim = QINF;
                if (au) { elo = -QINF; ehi = 0.0; } else { elo = 0.0; ehi = QINF; }
            }
            STAMP(4);
        } else {
            int r;
            if (__popcll(bal) == 1) r = __ffsll((long long)bal) - 1;
            else {
                const bool cand = dd <= dmax;
                double ag = cand ? fabs(gdir) : -1.0;
                if (cand && rowvar == VTH) ag = QINF;
                const double bestg = wave_max_f64(ag);
                r = wave_first(cand && ag == bestg);
            }
            r = uni(r);
            STAMP(2);   // ratio test + row choice
            // ---- pivot row r -> svrow (raw): the 16 lanes that hold it pick the register under a uniform tree
            const int rq = r & 3;
            rsel = ((r >> 4) << 2) | ((r >> 2) & 3);      // leaf = Ib*4 + g
            {
                const double pr0 = SJ0[rsel], pr1 = SJ1[rsel];          // uniform dynamic register index
                if (lq == rq) { svrow[lc] = pr0; svrow[16 + lc] = pr1; }
            }
            // the row's extra-column entry rides along as "column 32", and the entry of the pivot column itself is
            // replaced by -1 so that row * inv carries -inv there (what the exchange needs) without any select
            if (l == r) { svrow[XC] = tcol; svrow[c] = -1.0; }      // (one lane, program order after the owners' stores above)
            double step = readlane_f64(dd, r);
            if (step < 0.0) step = 0.0;
            const double leave_val = readlane_f64(tb, r);
            const double rcr = readlane_f64(rc, r);
            inv = sneg ? -rcr : rcr;
            const double delta = sneg ? -step : step;
            const int vl = readlane_i32(rowvar, r);
            const double enter_val = readlane_f64(nbval, c) + delta;
            wave_sync();
            {
                double pa = svrow[lc], pb = svrow[16 + lc], px = svrow[XC];
                asm volatile("" : "+v"(pa), "+v"(pb), "+v"(px));
                v0 = pa * inv; v1 = pb * inv;
                // extra column and values (lane l <-> row l): the same exchange formulas, the pivot column's own
                // entries start from 0 when it is the extra column
                const double vx = px * inv;
                const double tc0 = (c == XC) ? 0.0 : tcol;
                double xbn = fma(delta, cm, xb);
                double tcn = fma(-cm, vx, tc0);
                if (l == r) { xbn = enter_val; tcn = -vx; }
                xb = xbn; tcol = tcn;
                // the lanes of tile column c & 15 / of lane group rq, as scalar shifts (no compare)
                mcol = 0x0001000100010001ull << (c & 15); mrow = 0xFFFFull << (16 * rq);
                cx = c;
            }
            STAMP(3);   // pivot row through LDS
            rW = r; veW = readlane_i32(colvar, c); eloW = elo; ehiW = ehi;      // row r now holds the entering variable
            cW = c; vlW = vl; nbW = leave_val;                                  // column c the leaving one
            pivots++;
            if (vl == VTH) { status = QPN_SUCCESS; stop = true; }
            else {
            int vn;
            {
                const int k = vl < NBP ? vl : vl - NBP;
                const int cls = readlane_i32(clsv, k);
                int au = readlane_i32(satv, k);
                if (vl < NBP) {
                    // the bounded variable p_k left at a bound -- the upper one iff row r was a `hi` ratio --:
                    // its multiplier d_k enters from 0
                    au = uni(__double2hiint(rcr)) >= 0 ? 1 : 0; kW = k; auW = au;      // (sign bit of the pivot, |pivot| > ptol: gdir_r > 0 <=> left at hi)
                    vn = NBP + k;
                    sneg = au != 0;
                    self_lim = QINF;
                    if (cls == 2) { elo = 0.0; ehi = 0.0; }
                    else if (au) { elo = -QINF; ehi = 0.0; }
                    else { elo = 0.0; ehi = QINF; }
                } else {
                    // the multiplier d_k left at 0: p_k enters, moving off the bound it rests at
                    vn = k;
                    sneg = au != 0;
                    self_lim = readlane_f64(rngv, k);           // +inf for a free pair
                    if (cls == 2) sneg = false;
                    elo = readlane_f64(lo0, k); ehi = readlane_f64(hi0, k);
                }
            }
            // (colvar still holds the entering id veW at column c here -- the write-back is below --, so that
            // column must not match; the new id there, vl, is never its own complement vn)
            cnext = (vn == veW) ? -1 : col_of(vn);
            if (cnext < 0) { status = QPN_FAILURE; stop = true; }
            }
            STAMP(4);
        }
        // the exchange's column operands: requested here so that the LDS trip overlaps the write-backs below
        const d4 ua = *reinterpret_cast<const d4 *>(sucol + lq * 8);
        const d4 ub = *reinterpret_cast<const d4 *>(sucol + lq * 8 + 4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- write-back of the single-lane bookkeeping updates
        // (ONE asm block for the nine v_writelane_b32: M0 -- the lane select, gfx9 allows one SGPR per VALU instruction -- is
        // saved and restored once instead of six times, and the four lane selects follow each other without the
        // compiler's copies in between: ~20 instructions less on the pivot's dependent chain)
        {
            const int s_rW = uni(rW), s_cW = uni(cW), s_cc = uni(c), s_kW = uni(kW);
            const int s_ve = uni(veW), s_vl = uni(vlW), s_au = uni(auW);
            const int s_el = uni(__double2loint(eloW)), s_eh = uni(__double2hiint(eloW));
            const int s_fl = uni(__double2loint(ehiW)), s_fh = uni(__double2hiint(ehiW));
            const int s_nl = uni(__double2loint(nbW)), s_nh = uni(__double2hiint(nbW));
            int lol = __double2loint(lo), loh = __double2hiint(lo), hil = __double2loint(hi), hih = __double2hiint(hi);
            int nbl = __double2loint(nbval), nbh = __double2hiint(nbval);
            int m0save;
            asm("s_mov_b32 %[sv], m0\n\t"
                "s_mov_b32 m0, %[lr]\n\ts_nop 3\n\t"
                "v_writelane_b32 %[rowvar], %[ve], m0\n\t"
                "v_writelane_b32 %[lol], %[el], m0\n\tv_writelane_b32 %[loh], %[eh], m0\n\t"
                "v_writelane_b32 %[hil], %[fl], m0\n\tv_writelane_b32 %[hih], %[fh], m0\n\t"
                "s_mov_b32 m0, %[lc]\n\ts_nop 0\n\t"
                "v_writelane_b32 %[colvar], %[vl], m0\n\t"
                "s_mov_b32 m0, %[cc]\n\ts_nop 0\n\t"
                "v_writelane_b32 %[nbl], %[nl], m0\n\tv_writelane_b32 %[nbh], %[nh], m0\n\t"
                "s_mov_b32 m0, %[lk]\n\ts_nop 0\n\t"
                "v_writelane_b32 %[satv], %[au], m0\n\t"
                "s_mov_b32 m0, %[sv]"
                : [rowvar] "+v"(rowvar), [lol] "+v"(lol), [loh] "+v"(loh), [hil] "+v"(hil), [hih] "+v"(hih), [colvar] "+v"(colvar),
                  [nbl] "+v"(nbl), [nbh] "+v"(nbh), [satv] "+v"(satv), [sv] "=&s"(m0save)
                : [lr] "s"(s_rW), [lc] "s"(s_cW), [cc] "s"(s_cc), [lk] "s"(s_kW), [ve] "s"(s_ve), [vl] "s"(s_vl), [au] "s"(s_au),
                  [el] "s"(s_el), [eh] "s"(s_eh), [fl] "s"(s_fl), [fh] "s"(s_fh), [nl] "s"(s_nl), [nh] "s"(s_nh));
            lo = __hiloint2double(loh, lol); hi = __hiloint2double(hih, hil); nbval = __hiloint2double(nbh, nbl);
        }
        if (stop) break;
        // ---- the exchange on the 32 x 32 dictionary (see the header of this stage)
        {
            // (row r becomes -v: a multiplication by -1.0 inside the block, exact, instead of two negated copies kept ready;
            //  1 / pivot goes in as the scalar it is)
            const double inv_s = udbl(inv);
            cx = uni(cx); rsel = uni(rsel);
            if constexpr (HALF16) {
                // n = m = 16: pairs 16.. are padding -- the other three tiles stay zero (u = 0 in rows 16.., v = 0 in columns 16..)
                asm volatile(
                    "v_fma_f64 %[s000], -%[u0], %[v0], %[s000]\n\tv_fma_f64 %[s001], -%[u1], %[v0], %[s001]\n\t"
                    "v_fma_f64 %[s002], -%[u2], %[v0], %[s002]\n\tv_fma_f64 %[s003], -%[u3], %[v0], %[s003]\n\t"
                    "s_cmp_gt_i32 %[c], 31\n\ts_cbranch_scc1 1f\n\t"
                    "s_mov_b64 exec, %[mcol]\n\t"
                    "v_mul_f64 %[s000], %[u0], %[iv]\n\tv_mul_f64 %[s001], %[u1], %[iv]\n\t"
                    "v_mul_f64 %[s002], %[u2], %[iv]\n\tv_mul_f64 %[s003], %[u3], %[iv]\n"
                    "1:\n\t"
                    "s_mov_b64 exec, %[mrow]\n\t"
                    "s_cmp_gt_i32 %[rs], 1\n\ts_cbranch_scc1 3f\n\t"
                    "s_cmp_eq_u32 %[rs], 0\n\ts_cbranch_scc0 2f\n\t"
                    "v_mul_f64 %[s000], %[v0], -1.0\n\ts_branch 5f\n"
                    "2:\n\tv_mul_f64 %[s001], %[v0], -1.0\n\ts_branch 5f\n"
                    "3:\n\ts_cmp_eq_u32 %[rs], 2\n\ts_cbranch_scc0 4f\n\t"
                    "v_mul_f64 %[s002], %[v0], -1.0\n\ts_branch 5f\n"
                    "4:\n\tv_mul_f64 %[s003], %[v0], -1.0\n"
                    "5:\n\t"
                    "s_mov_b64 exec, -1"
                    : [s000] "+v"(SD(0, 0, 0)), [s001] "+v"(SD(0, 0, 1)), [s002] "+v"(SD(0, 0, 2)), [s003] "+v"(SD(0, 0, 3))
                    : [u0] "v"(ua[0]), [u1] "v"(ua[1]), [u2] "v"(ua[2]), [u3] "v"(ua[3]), [v0] "v"(v0), [iv] "s"(inv_s),
                      [mcol] "s"(mcol), [mrow] "s"(mrow), [c] "s"(cx), [rs] "s"(rsel)
                    : "scc");
            } else
            asm volatile(
                "v_fma_f64 %[s000], -%[u0], %[v0], %[s000]\n\tv_fma_f64 %[s001], -%[u1], %[v0], %[s001]\n\t"
                "v_fma_f64 %[s002], -%[u2], %[v0], %[s002]\n\tv_fma_f64 %[s003], -%[u3], %[v0], %[s003]\n\t"
                "v_fma_f64 %[s010], -%[u0], %[v1], %[s010]\n\tv_fma_f64 %[s011], -%[u1], %[v1], %[s011]\n\t"
                "v_fma_f64 %[s012], -%[u2], %[v1], %[s012]\n\tv_fma_f64 %[s013], -%[u3], %[v1], %[s013]\n\t"
                "v_fma_f64 %[s100], -%[u4], %[v0], %[s100]\n\tv_fma_f64 %[s101], -%[u5], %[v0], %[s101]\n\t"
                "v_fma_f64 %[s102], -%[u6], %[v0], %[s102]\n\tv_fma_f64 %[s103], -%[u7], %[v0], %[s103]\n\t"
                "v_fma_f64 %[s110], -%[u4], %[v1], %[s110]\n\tv_fma_f64 %[s111], -%[u5], %[v1], %[s111]\n\t"
                "v_fma_f64 %[s112], -%[u6], %[v1], %[s112]\n\tv_fma_f64 %[s113], -%[u7], %[v1], %[s113]\n\t"
                // column c: T[i][c] = u_i * inv on the 4 lanes that own it (none when c is the extra column)
                "s_cmp_gt_i32 %[c], 31\n\ts_cbranch_scc1 .Lqx_row%=\n\t"
                "s_mov_b64 exec, %[mcol]\n\t"
                "s_cmp_gt_i32 %[c], 15\n\ts_cbranch_scc1 .Lqx_col1%=\n\t"
                "v_mul_f64 %[s000], %[u0], %[iv]\n\tv_mul_f64 %[s001], %[u1], %[iv]\n\t"
                "v_mul_f64 %[s002], %[u2], %[iv]\n\tv_mul_f64 %[s003], %[u3], %[iv]\n\t"
                "v_mul_f64 %[s100], %[u4], %[iv]\n\tv_mul_f64 %[s101], %[u5], %[iv]\n\t"
                "v_mul_f64 %[s102], %[u6], %[iv]\n\tv_mul_f64 %[s103], %[u7], %[iv]\n\t"
                "s_branch .Lqx_row%=\n"
                ".Lqx_col1%=:\n\t"
                "v_mul_f64 %[s010], %[u0], %[iv]\n\tv_mul_f64 %[s011], %[u1], %[iv]\n\t"
                "v_mul_f64 %[s012], %[u2], %[iv]\n\tv_mul_f64 %[s013], %[u3], %[iv]\n\t"
                "v_mul_f64 %[s110], %[u4], %[iv]\n\tv_mul_f64 %[s111], %[u5], %[iv]\n\t"
                "v_mul_f64 %[s112], %[u6], %[iv]\n\tv_mul_f64 %[s113], %[u7], %[iv]\n"
                // row r: T[r][j] = -v_j on the 16 lanes that own it (v carries -inv at column c)
                ".Lqx_row%=:\n\t"
                "s_mov_b64 exec, %[mrow]\n\t"
                "s_cmp_gt_i32 %[rs], 3\n\ts_cbranch_scc1 .Lqx_r4%=\n\t"
                "s_cmp_gt_i32 %[rs], 1\n\ts_cbranch_scc1 .Lqx_r2%=\n\t"
                "s_cmp_eq_u32 %[rs], 0\n\ts_cbranch_scc0 .Lqx_r1%=\n\t"
                "v_mul_f64 %[s000], %[v0], -1.0\n\tv_mul_f64 %[s010], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r1%=:\n\tv_mul_f64 %[s001], %[v0], -1.0\n\tv_mul_f64 %[s011], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r2%=:\n\ts_cmp_eq_u32 %[rs], 2\n\ts_cbranch_scc0 .Lqx_r3%=\n\t"
                "v_mul_f64 %[s002], %[v0], -1.0\n\tv_mul_f64 %[s012], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r3%=:\n\tv_mul_f64 %[s003], %[v0], -1.0\n\tv_mul_f64 %[s013], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r4%=:\n\ts_cmp_gt_i32 %[rs], 5\n\ts_cbranch_scc1 .Lqx_r6%=\n\t"
                "s_cmp_eq_u32 %[rs], 4\n\ts_cbranch_scc0 .Lqx_r5%=\n\t"
                "v_mul_f64 %[s100], %[v0], -1.0\n\tv_mul_f64 %[s110], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r5%=:\n\tv_mul_f64 %[s101], %[v0], -1.0\n\tv_mul_f64 %[s111], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r6%=:\n\ts_cmp_eq_u32 %[rs], 6\n\ts_cbranch_scc0 .Lqx_r7%=\n\t"
                "v_mul_f64 %[s102], %[v0], -1.0\n\tv_mul_f64 %[s112], %[v1], -1.0\n\ts_branch .Lqx_end%=\n"
                ".Lqx_r7%=:\n\tv_mul_f64 %[s103], %[v0], -1.0\n\tv_mul_f64 %[s113], %[v1], -1.0\n"
                ".Lqx_end%=:\n\t"
                "s_mov_b64 exec, -1"
                : [s000] "+v"(SD(0, 0, 0)), [s001] "+v"(SD(0, 0, 1)), [s002] "+v"(SD(0, 0, 2)), [s003] "+v"(SD(0, 0, 3)),
                  [s010] "+v"(SD(0, 1, 0)), [s011] "+v"(SD(0, 1, 1)), [s012] "+v"(SD(0, 1, 2)), [s013] "+v"(SD(0, 1, 3)),
                  [s100] "+v"(SD(1, 0, 0)), [s101] "+v"(SD(1, 0, 1)), [s102] "+v"(SD(1, 0, 2)), [s103] "+v"(SD(1, 0, 3)),
                  [s110] "+v"(SD(1, 1, 0)), [s111] "+v"(SD(1, 1, 1)), [s112] "+v"(SD(1, 1, 2)), [s113] "+v"(SD(1, 1, 3))
                : [u0] "v"(ua[0]), [u1] "v"(ua[1]), [u2] "v"(ua[2]), [u3] "v"(ua[3]), [u4] "v"(ub[0]), [u5] "v"(ub[1]),
                  [u6] "v"(ub[2]), [u7] "v"(ub[3]), [v0] "v"(v0), [v1] "v"(v1), [iv] "s"(inv_s),
                  [mcol] "s"(mcol), [mrow] "s"(mrow), [c] "s"(cx), [rs] "s"(rsel)
                : "scc");
        }
        c = cnext;
        wave_sync();
        STAMP(3);
    }
    STAMP(4);   // Lemke: exit paths
    // ---- everything below reads its kernel arguments AFRESH from the kernarg segment (through an opaque
    // pointer), so that output pointers, tolerances and record bases are not kept in SGPRs -- and spilled to
    // VGPR lanes -- across the pivot loops above
    typedef const AviBatchArgs __attribute__((address_space(4))) *kargs_t;
    kargs_t kp = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    AviBatchArgs ae{};
    ae.z = kp->z; ae.status = kp->status; ae.resid = kp->resid; ae.pivots = kp->pivots; ae.active = kp->active;
    ae.check_tol = kp->check_tol; ae.comp_tol = kp->comp_tol; ae.x = kp->x; ae.stride_x = kp->stride_x;
#ifdef QPN_STAMPS
    ae.stamps = kp->stamps;
#endif
    if constexpr (NODES) { ae.nd.Qd = kp->nd.Qd; ae.nd.l = kp->nd.l; ae.nd.u = kp->nd.u; }
    else { ae.M = kp->M; ae.strideM = kp->strideM; ae.q = kp->q; ae.l = kp->l; ae.u = kp->u; ae.kind = kp->kind; ae.stride_kind = kp->stride_kind; }
    const double *Qe_ = NODES ? ae.nd.Qd + (size_t)b * nn * nn : nullptr;
    const double *Mge = NODES ? nullptr : ae.M + (size_t)b * (size_t)ae.strideM;
    auto row_bounds_e = [&](double &lk_, double &uk_, int &gk_) {
        lk_ = 0.0; uk_ = 0.0; gk_ = 0;
        if constexpr (NODES) {
            if (act) {
                if (l < nn) { lk_ = -QINF; uk_ = QINF; }
                else { lk_ = ae.nd.l[(size_t)b * nm + (l - nn)]; uk_ = ae.nd.u[(size_t)b * nm + (l - nn)]; gk_ = 1; }
            }
        } else {
            lk_ = act ? ae.l[vo + l] : 0.0; uk_ = act ? ae.u[vo + l] : 0.0;
            gk_ = (act && ae.kind) ? (int)ae.kind[(size_t)b * (size_t)ae.stride_kind + l] : 0;
        }
    };
    // ---- post-check operands requested first: the bounds of item row l and (n = 32) this lane's 32 entries of
    // Qd, so that the round trip hides behind the read-back below instead of four short ones in the check loop
    double lk, uk; int gk;
    row_bounds_e(lk, uk, gk);
    double mq32[32];
    // ---- read back: lambda_k, then x = -(W lambda + h) ---------------------------------------------------
    // lane coordinates recomputed from the lane id here (two VALU instructions): kept alive across the pivot loops they
    // are what the register allocator spills to scratch
    int lE_ = l; asm volatile("" : "+v"(lE_));
    const int lcE = lE_ & 15, lqE = lE_ >> 4;
    wave_sync();
    if (actb) sval[rowvar] = xb;
    if (l <= XC) sval[colvar] = nbval;
    wave_sync();
    const double lam0 = sval[NBP + lcE], lam1 = sval[NBP + 16 + lcE];     // lambda of this lane's two columns
    {
        // (W lambda)_row = sum over the 16 lanes of a DPP row of this lane's two-column partial, for the 8
        // rows j = 4 Ib + g a lane holds.  Folded butterfly: at each of the first three stages a lane gives
        // half of its values to its partner and adds the partner's other half, so 8 -> 4 -> 2 -> 1 values
        // (7 exchanges instead of 8 x 4); lane bits 0..2 of lc then name the row, bit 3 is summed last.
        double p0 = fma(TL(0, 2)[0], lam0, TL(0, 3)[0] * lam1), p1 = fma(TL(0, 2)[1], lam0, TL(0, 3)[1] * lam1);
        double p2 = fma(TL(0, 2)[2], lam0, TL(0, 3)[2] * lam1), p3 = fma(TL(0, 2)[3], lam0, TL(0, 3)[3] * lam1);
        double p4 = fma(TL(1, 2)[0], lam0, TL(1, 3)[0] * lam1), p5 = fma(TL(1, 2)[1], lam0, TL(1, 3)[1] * lam1);
        double p6 = fma(TL(1, 2)[2], lam0, TL(1, 3)[2] * lam1), p7 = fma(TL(1, 2)[3], lam0, TL(1, 3)[3] * lam1);
        // (the W tiles are dead from here on: their 32 registers take the 32 entries of Qd this lane needs for the
        // post-check, requested now so that the round trip hides behind the reduction and its LDS hops)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NODES && FULL32) {
            // (x rows: this lane's row of Qd from global memory; constraint rows: this lane's row of Ad from the LDS block buffer
            //  -- two exec-masked batches into the SAME registers, so that the post-check is ONE fma chain over the x columns)
            if (l < 32) {
                const double *qc = Qe_ + l;
#pragma unroll
                for (int j = 0; j < 32; ++j) mq32[j] = qc[(size_t)j * 32];
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) mq32[j] = sA[j * SAS + (l - 32)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool b0 = (lcE & 1) != 0, b1 = (lcE & 2) != 0, b2 = (lcE & 4) != 0;
        // stage 1 (partner lc ^ 1): keep rows with bit0 == b0
        const double q0 = (b0 ? p1 : p0) + dpp_f64<0xB1>(b0 ? p0 : p1), q1 = (b0 ? p3 : p2) + dpp_f64<0xB1>(b0 ? p2 : p3);
        const double q2 = (b0 ? p5 : p4) + dpp_f64<0xB1>(b0 ? p4 : p5), q3 = (b0 ? p7 : p6) + dpp_f64<0xB1>(b0 ? p6 : p7);
        // stage 2 (partner lc ^ 2): q_i is row 2 i + b0; keep rows with bit1 == b1
        const double r0 = (b1 ? q1 : q0) + dpp_f64<0x4E>(b1 ? q0 : q1), r1 = (b1 ? q3 : q2) + dpp_f64<0x4E>(b1 ? q2 : q3);
        // stages 3, 4 (partners lc ^ 4, lc ^ 8) go through ds_swizzle: the LDS crossbar, not the VALU
        auto swz = [](double v, auto pat) -> double {
            const int lo_ = __builtin_amdgcn_ds_swizzle(__double2loint(v), decltype(pat)::value);
            const int hi_ = __builtin_amdgcn_ds_swizzle(__double2hiint(v), decltype(pat)::value);
            return __hiloint2double(hi_, lo_);
        };
        double t0 = (b2 ? r1 : r0) + swz(b2 ? r0 : r1, std::integral_constant<int, 0x101F>{});   // keep rows with bit2 == b2
        t0 += swz(t0, std::integral_constant<int, 0x201F>{});
        // the lane now holds the full sum for row j = b0 + 2 b1 + 4 b2 of its own lc bits
        const int jrow = lcE & 7;
        if (lcE < 8) sz[16 * (jrow >> 2) + 4 * (jrow & 3) + lqE] = t0;
    }
    wave_sync();
    // item order: rows < n are x, rows n.. are lambda
    double zk = 0.0;
    if (act) zk = l < n ? (NODES ? sz[l] - kx : -(sz[l] + kx)) : sval[NBP + (l - n)];      // node path: sz = W~ lambda = -W lambda
    wave_sync();
    if (act) sz[l] = zk;
    wave_sync();

    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -------------------------------
    double rk;
    if constexpr (NODES) {
        // r = q + M z, item columns in ascending order (finite blocks: a zero z_j contributes exactly
        // nothing).  x rows read Qd column-wise from global memory (lane <-> row: coalesced, L2 / Infinity
        // Cache hits) and their -Ad' part from LDS; constraint rows read Ad from LDS.  Every lane runs BOTH
        // fma chains (one is garbage, on in-range addresses) and keeps its own at the end: two v_fma_f64
        // per column instead of a compare and four selects.
        const bool isx = l < nn;
        const int ls = act ? (isx ? l : l - nn) : 0;
        const double *qcol = Qe_ + (isx ? ls : 0);
        const int aoff = isx ? 0 : ls, roff = (isx ? ls : 0) * SAS;   // column of Ad (constraint rows) / row of Ad' (x rows)
        double rq = SQ(l), ra = rq;
        int j = 0;
        if constexpr (FULL32) {
            // the x columns: ONE chain for every lane -- its operands (a row of Qd or a row of Ad) were fetched into mq32 by role
            // (two accumulators: the chain is latency-bound, not issue-bound)
            double rq2 = 0.0;
#pragma unroll
            for (int jj = 0; jj < 32; jj += 2) { rq = fma(mq32[jj], sz[jj], rq); rq2 = fma(mq32[jj + 1], sz[jj + 1], rq2); }
            rq += rq2;
            ra = rq;
            j = 32;
        }
        for (; j + 8 <= nn; j += 8) {
            double mq[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mq[q8] = qcol[(size_t)(j + q8) * nn];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                const double zj = sz[j + q8];
                rq = fma(mq[q8], zj, rq);
                ra = fma(sA[(j + q8) * SAS + aoff], zj, ra);
            }
        }
        for (; j < nn; ++j) {
            const double zj = sz[j];
            rq = fma(qcol[(size_t)j * nn], zj, rq);
            ra = fma(sA[j * SAS + aoff], zj, ra);
        }
        if constexpr (FULL32) {
            const int rof2 = (l & 31) * SAS + ((l >> 5) << 4);
            const double *const szl = sz + 32 + ((l >> 5) << 4);
            double rl1 = 0.0, rl2 = 0.0;                                            // columns of lambda: -A' (x rows), two chains
#pragma unroll 8
            for (int k = 0; k < 16; k += 2) { rl1 = fma(-sA[rof2 + k], szl[k], rl1); rl2 = fma(-sA[rof2 + k + 1], szl[k + 1], rl2); }
            rq += sum_halves(rl1 + rl2);
        } else {
#pragma unroll 8
        for (int k = 0; k < nm; ++k) rq = fma(-sA[roff + k], sz[nn + k], rq);      // columns of lambda: -A' (x rows)
        }
        rk = isx ? rq : ra;
    } else {
        rk = act ? ae.q[vo + l] : 0.0;
        int j = 0;
        if constexpr (FULL32) {
            // N = 64: two batches of 32 columns, each requested at once (the W tiles are dead: their registers hold the batch),
            // two chains per batch
            double rk2 = 0.0;
#pragma unroll 1
            for (; j < 64; j += 32) {
                double mv[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) mv[q] = Mge[(size_t)(j + q) * 64 + l];
#pragma unroll
                for (int q = 0; q < 32; q += 2) {
                    const double za = sz[j + q], zb = sz[j + q + 1];
                    rk = (za != 0.0) ? fma(mv[q], za, rk) : rk;
                    rk2 = (zb != 0.0) ? fma(mv[q + 1], zb, rk2) : rk2;
                }
            }
            rk += rk2;
        }
#pragma unroll 1
        for (; j + 8 <= N; j += 8) {
            double mv[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) mv[q8] = act ? Mge[(size_t)(j + q8) * N + l] : 0.0;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { const double zj = sz[j + q8]; rk = (zj != 0.0) ? fma(mv[q8], zj, rk) : rk; }
        }
        for (; j < N; ++j) { const double zj = sz[j]; if (zj != 0.0 && act) rk = fma(Mge[(size_t)j * N + l], zj, rk); }
    }
    const double pp = gk ? rk : zk, dv = gk ? zk : rk;
    int bad = 0;
    double nres = 0.0;
    unsigned mask = 0;
    if (act) {
        const double tol = ae.check_tol;
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
        const double ct = ae.comp_tol;
        // x == y, or both finite and within ct: an infinite operand makes |x - y| infinite (or NaN for inf - inf, which the
        // first clause has already taken), so the finiteness tests of the scalar statement are implied
        auto approx = [&](double x, double y) { return (x == y) | (fabs(x - y) <= ct); };
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
        ae.z[vo + l] = zk;
        if (ae.x && l < n) {                                                    // primal block -> the caller's iterate
            const size_t xo = (size_t)b * (size_t)ae.stride_x + l;
            ae.x[xo] = zk;
            const int nmir = kp->n_mirror;                                      // ... and its replicas on the peer GPUs
            for (int k = 0; k < nmir; ++k) kp->mirror[k][xo] = zk;
        }
        if (ae.active) ae.active[vo + l] = (uint8_t)mask;
    }
    if (l == 0) {
        ae.status[b] = status;
        if (ae.resid) ae.resid[b] = nres;
        if (ae.pivots) ae.pivots[b] = pivots;
        if constexpr (NODES) {
            int32_t *const sk = kp->sched_key;                                  // this node's smoothed pivot count
            if (sk) { const int k0 = sk[b]; sk[b] = k0 > 0 ? k0 - (k0 >> 5) + pivots : 32 * pivots; }
        }
    }
    STAMP(5);   // read-back + post-check + stores
#ifdef QPN_STAMPS
    if (ae.stamps && l == 0) {
        // wall-clock start / end of this block (s_memrealtime: one 100 MHz counter for the whole device)
        stamp_acc[7] = (stamp_rt0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull);
        for (int i = 0; i < 8; ++i) ae.stamps[(size_t)b * 8 + i] = stamp_acc[i];
    }
#endif
}

} // namespace

hipError_t qpn_launch_avi_solve_schur(const AviBatchArgs &a, double *dbgS, double *dbgc, double *dbgW,
                                      double *dbgh, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    SchurDebug d{dbgS, dbgc, dbgW, dbgh};
    if (a.N == 64 && !dbgS)
        hipLaunchKernelGGL((avi_solve_schur<false, 0, 32>), dim3((unsigned)a.batch), dim3(WAVE), 0, stream, a, d);
    else
        hipLaunchKernelGGL((avi_solve_schur<false, 0, 0>), dim3((unsigned)a.batch), dim3(WAVE), 0, stream, a, d);
    return hipGetLastError();
}

hipError_t qpn_launch_avi_solve_schur_nodes(const AviBatchArgs &a, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    SchurDebug d{nullptr, nullptr, nullptr, nullptr};
#ifdef QPN_DIAG
    // diagnostic builds only: extra dynamic LDS per block lowers the occupancy (occupancy-sensitivity experiments)
    static const unsigned pad = [] { const char *e = QPN_DEV_ENV("QPN_DEBUG_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
#else
    const unsigned pad = 0;
#endif
    // resident wavefronts of this kernel: 16 per CU (LDS- and VGPR-bound, see the kernel header)
    static int resident[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (resident[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        resident[dev] = 16 * cus;
    }
    static const bool no_stagger = [] { const char *e = QPN_DEV_ENV("QPN_NO_STAGGER"); return e && e[0] == '1'; }();   // A/B switch
    // a partial round has no burst to spread; other CU counts (partitioned modes) run without the stagger
    const bool stag = !no_stagger && resident[dev] == kResidentMI355X && a.batch > kResidentMI355X;
    const bool full = a.nd.n == 32 && a.nd.m == 32, half = a.nd.n == 16 && a.nd.m == 16;
    const dim3 grid((unsigned)a.batch), block(WAVE);
    if (full && a.nd.sym) {
        if (stag) hipLaunchKernelGGL((avi_solve_schur<true, kResidentMI355X, 32, true>), grid, block, pad, stream, a, d);
        else hipLaunchKernelGGL((avi_solve_schur<true, 0, 32, true>), grid, block, pad, stream, a, d);
        return hipGetLastError();
    }
    if (stag && full) hipLaunchKernelGGL((avi_solve_schur<true, kResidentMI355X, 32>), grid, block, pad, stream, a, d);
    else if (stag && half) hipLaunchKernelGGL((avi_solve_schur<true, kResidentMI355X, 16>), grid, block, pad, stream, a, d);
    else if (stag) hipLaunchKernelGGL((avi_solve_schur<true, kResidentMI355X, 0>), grid, block, pad, stream, a, d);
    else if (full) hipLaunchKernelGGL((avi_solve_schur<true, 0, 32>), grid, block, pad, stream, a, d);
    else if (half) hipLaunchKernelGGL((avi_solve_schur<true, 0, 16>), grid, block, pad, stream, a, d);
    else hipLaunchKernelGGL((avi_solve_schur<true, 0, 0>), grid, block, pad, stream, a, d);
    return hipGetLastError();
}
