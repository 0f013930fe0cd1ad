// Fused node-AVI solve for node records with 33 <= max(n, m) <= 48 on ONE wavefront per node (src/avi.jl:205-251, :305-377 assembly;
// :63-77 solve; :148-156 post-check; src/avi_solutions.jl:511-562 masks) -- the algorithm and the tile layout of the 32-class kernel
// (qpn_avi_schur.hip: crash of the x block as rank-4 block pivots on the fp64 matrix cores, Lemke with the Harris two-pass ratio
// test on the Schur dictionary in registers, post-check on the original blocks), with THREE 16 x 16 tiles a side instead of two.
// 48 rows fit the 64 lanes of a wavefront (one row per lane in the ratio test), and at two wavefronts per SIMD a wave may use 256
// VGPRs: 18 tiles of [H | C~] in the crash, 9 + 9 tiles (dictionary + W~) in the Lemke phase.  The fused workgroup kernel of the
// 33-64 class (qpn_avi_schur_wg.hip) spends 12 K VALU instructions per node at n = m = 48 on three waves that each factor the
// pivot block, keep their own bookkeeping and meet at two barriers per pivot; this kernel has no barrier at all.
// Sizes inside the class are padded to 48: identity rows in H, zero rows / columns elsewhere.  Items the crash cannot take
// (an equality row, a block pivot below the threshold) are flagged status = -1 for the general path, as in the other kernels.
#include "qpn_internal.h"

#define QINF __builtin_huge_val()

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)
#define MFMA_NEGA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 1)      // D = C - A B (gfx950 NEG bits)

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double rcp64_(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
#ifdef QPN_STAMPS
#define STAMP(slot) do { unsigned long long now__ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); stamp_acc[slot] += now__ - stamp_last; stamp_last = now__; } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

constexpr int T3 = 3, NP = 48, SAS = 49;       // tiles a side, padded size, column stride of the LDS block buffer (odd: conflict-free)

__global__ __launch_bounds__(64, 2) void avi_solve_schur48(AviBatchArgs a)
{
    const int l = threadIdx.x, lc = l & 15, lq = l >> 4;
    int b = blockIdx.x;
    if (a.order) { b = a.order[blockIdx.x]; if ((unsigned)b >= (unsigned)a.batch) return; }
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    auto decline = [&]() {
        if (l == 0) { a.status[b] = -1; if (a.decl_count) atomicAdd(a.decl_count, 1); }
    };
    if (!(n >= 1 && n <= NP && m >= 0 && m <= NP)) { decline(); return; }
#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif

    // block buffer: Qd while the H tiles are built, then Ad ([column of x][constraint row]) through the S product, then W~ for the
    // read-back, then the solution in item order.  18 816 + 1 664 B = 20 480 B: EIGHT wavefronts per CU (two per SIMD); q never
    // goes to LDS (three values per lane, in registers).
    __shared__ __attribute__((aligned(32))) double sA[NP * SAS];
    __shared__ __attribute__((aligned(32))) double sbuf[208];
    double *const sU = sbuf;                    // Stage A: pivot columns, [48][4]
    double *const sP = sbuf + 192;              // Stage A: the raw 4 x 4 pivot block
    double *const sh = sbuf;                    // h for c = b - A h [48]
    double *const sucol = sbuf;                 // Stage B: entering column, permuted [4][12]
    double *const svrow = sbuf + 64;            // Stage B: pivot row [49]
    double *const sval = sbuf;                  // read-back: values by variable id [97]
    // the solution in item order lies over the first two columns of the block buffer (entries 48 of a column are skipped)
#define SZ(i) sA[(i) + ((i) >= NP ? 1 : 0)]

    const double *Q_ = a.nd.Qd + (size_t)b * n * n;
    const double *A_ = a.nd.Ad + (size_t)b * m * n;
    const double *R_ = a.nd.R + (size_t)b * n * np_;
    const double *B_ = a.nd.B + (size_t)b * m * np_;
    const double *w_ = a.nd.w + (size_t)b * (size_t)a.nd.stride_w;

    // ---- q = [qd + R w; B w]: item rows l and 64 + l (for the post-check) and row n + l (c of pair l), the p terms in ascending
    // order, eight in flight together
    auto qterm = [&](int it) -> double {
        const bool in = it < N, isx = it < n;
        const int its = in ? it : 0;
        const double *col = (isx || !in) ? R_ + (isx ? its : 0) : B_ + (its - n);
        const size_t cs = (isx || !in) ? (size_t)n : (size_t)m;
        double s = (in && isx) ? a.nd.qd[(size_t)b * n + its] : 0.0;
        for (int k0 = 0; k0 < np_; k0 += 8) {
            double rv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) rv[k] = col[(in && k0 + k < np_) ? (size_t)(k0 + k) * cs : 0];
#pragma unroll
            for (int k = 0; k < 8; ++k) s = (in && k0 + k < np_) ? fma(rv[k], w_[k0 + k < np_ ? k0 + k : 0], s) : s;
        }
        return s;
    };
    const double qv0 = qterm(l), qv1 = (N > 64) ? qterm(64 + l) : 0.0, qvc = (l < m) ? qterm(n + l) : 0.0;
    // ---- Qd -> LDS (zero-padded), column by column: lane <-> row, 16 columns in flight (all 48 + 48 columns of Qd and Ad in
    // flight at once shorten a lone node by 2 % and cost 4 % at 4 000 nodes: the resident waves' bursts collide)
    double mabs = 0.0;
#pragma unroll 1
    for (int c0 = 0; c0 < NP; c0 += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = c0 + t;
            const bool ok = c < n && l < n;
            const double x = Q_[ok ? (size_t)c * n + l : 0];
            v[t] = ok ? x : 0.0;
        }
        if (l < NP) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { sA[(c0 + t) * SAS + l] = v[t]; mabs = fmax(mabs, fabs(v[t])); }
        }
    }
    wsync();
    // the top half [H | C~] in the accumulator layout of v_mfma_f64_16x16x4_f64: tile (I, J), register g, lane (lq, lc) holds
    // row 16 I + 4 g + lq, column 16 J + lc
    d4 T[T3][2 * T3];
#pragma unroll
    for (int I = 0; I < T3; ++I)
#pragma unroll
        for (int J = 0; J < T3; ++J)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rr = 16 * I + 4 * g + lq, cc = 16 * J + lc;
                double v = sA[cc * SAS + rr];
                if (rr == cc && rr >= n) v = 1.0;                     // padded x rows: identity
                T[I][J][g] = v;
            }
    wsync();
    // ---- Ad -> LDS: sA[x column j][constraint row r] (zero-padded)
#pragma unroll 1
    for (int c0 = 0; c0 < NP; c0 += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = c0 + t;
            const bool ok = j < n && l < m;
            const double x = A_[ok ? (size_t)j * m + l : 0];
            v[t] = ok ? x : 0.0;
        }
        if (l < NP) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { sA[(c0 + t) * SAS + l] = v[t]; mabs = fmax(mabs, fabs(v[t])); }
        }
    }
    wsync();
    // the tiles hold -C = +Ad' (W~ = -W: S = D - A W = A W~, x = W~ lambda - h)
#pragma unroll
    for (int I = 0; I < T3; ++I)
#pragma unroll
        for (int J = 0; J < T3; ++J)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rr = 16 * I + 4 * g + lq, ck = 16 * J + lc;
                T[I][T3 + J][g] = sA[rr * SAS + ck];                   // zero outside n x m already
            }
    STAMP(0);   // loads + tiles
    // extra column: g = q of the x rows, lane l <-> row l (lanes >= 48 idle)
    double kx = (l < n) ? qv0 : 0.0;
    const double mscale = wave_max_f64(mabs);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);

    // ---- Stage A: 12 rank-4 block pivots of the top half on the matrix cores (pivot rows carry P - I in U, so that the update
    // T -= (U P^-1) V turns them into P^-1 V themselves)
    bool fail = false;
#pragma unroll
    for (int KB = 0; KB < 12; ++KB) {
        if (!fail && 4 * KB < n) {            // a block of padded rows is an identity pivot: nothing moves
            const int JP = KB >> 2, GP = KB & 3, p0 = 4 * KB;
            const int kcol = lc - 4 * GP;
            if (kcol >= 0 && kcol < 4) {
#pragma unroll
                for (int I = 0; I < T3; ++I)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int rr = 16 * I + 4 * g + lq;
                        double v = T[I][JP][g];
                        if (I == JP && g == GP) sP[lq * 4 + kcol] = v;                 // the pivot block itself, raw
                        if (I == JP && g == GP && lq == kcol) v -= 1.0;
                        sU[rr * 4 + kcol] = v;
                    }
            }
            wsync();
            double pm[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const d4 row = *reinterpret_cast<const d4 *>(sP + i * 4);
                pm[i][0] = row[0]; pm[i][1] = row[1]; pm[i][2] = row[2]; pm[i][3] = row[3];
            }
            const double x0 = readlane_f64(kx, p0), x1 = readlane_f64(kx, p0 + 1);
            const double x2 = readlane_f64(kx, p0 + 2), x3 = readlane_f64(kx, p0 + 3);
            bool okp = true;
            double rd[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                okp = okp && fabs(pm[s][s]) >= diag_thr;
                rd[s] = rcp64_(pm[s][s]);
#pragma unroll
                for (int i = s + 1; i < 4; ++i) {
                    const double f = pm[i][s] * rd[s];
                    pm[i][s] = f;
#pragma unroll
                    for (int j = s + 1; j < 4; ++j) pm[i][j] = fma(-f, pm[s][j], pm[i][j]);
                }
            }
            if (!ubool(okp)) { fail = true; }
            else {
                if (l < NP) {
                    const d4 ur = *reinterpret_cast<const d4 *>(sU + l * 4);
                    const double y0 = ur[0] * rd[0];
                    const double y1 = fma(-y0, pm[0][1], ur[1]) * rd[1];
                    const double y2 = fma(-y1, pm[1][2], fma(-y0, pm[0][2], ur[2])) * rd[2];
                    const double y3 = fma(-y2, pm[2][3], fma(-y1, pm[1][3], fma(-y0, pm[0][3], ur[3]))) * rd[3];
                    d4 up;
                    up[3] = y3;
                    up[2] = fma(-up[3], pm[3][2], y2);
                    up[1] = fma(-up[3], pm[3][1], fma(-up[2], pm[2][1], y1));
                    up[0] = fma(-up[3], pm[3][0], fma(-up[2], pm[2][0], fma(-up[1], pm[1][0], y0)));
                    kx -= fma(up[3], x3, fma(up[2], x2, fma(up[1], x1, up[0] * x0)));
                    *reinterpret_cast<d4 *>(sU + l * 4) = up;
                }
                wsync();
                double au[T3];
#pragma unroll
                for (int I = 0; I < T3; ++I) au[I] = sU[(16 * I + lc) * 4 + lq];
#pragma unroll
                for (int J = 0; J < 2 * T3; ++J) {
                    if (J >= JP) {                       // (columns left of the pivot tile are dead: H^-1 itself is never needed)
                        double vraw = T[JP][J][GP];
                        asm volatile("" : "+v"(vraw));
#pragma unroll
                        for (int I = 0; I < T3; ++I) T[I][J] = MFMA_NEGA(au[I], vraw, T[I][J]);
                    }
                }
                wsync();
            }
        }
    }
    if (fail) { decline(); return; }
    STAMP(1);   // stage A

    // ---- S = A W~ on the matrix cores (W~ = rows of x, an aligned group of 4 rows is a B operand), c = b - A h
    d4 SB[T3][T3];
    {
        const d4 z4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int Ib = 0; Ib < T3; ++Ib)
#pragma unroll
            for (int Jb = 0; Jb < T3; ++Jb) SB[Ib][Jb] = z4;
    }
    if (l < NP) sh[l] = kx;
    // bounds of pair l: requested here so that the round trip hides behind the MFMAs
    double lo = -QINF, hi = QINF;
    if (l < m) { lo = a.nd.l[(size_t)b * m + l]; hi = a.nd.u[(size_t)b * m + l]; }
#pragma unroll
    for (int I = 0; I < T3; ++I)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kk = 4 * I + g;
            if (4 * kk < n) {                 // rows of W~ beyond n are zero
                double ao[T3];
#pragma unroll
                for (int Ib = 0; Ib < T3; ++Ib) ao[Ib] = sA[(4 * kk + lq) * SAS + 16 * Ib + lc];
#pragma unroll
                for (int Ib = 0; Ib < T3; ++Ib)
#pragma unroll
                    for (int Jb = 0; Jb < T3; ++Jb) SB[Ib][Jb] = MFMA(ao[Ib], T[I][T3 + Jb][g], SB[Ib][Jb]);
            }
        }
    wsync();
    double xb = 0.0;
    {
        const bool lowr = l < m;
        const int ls = lowr ? l : 0;
        double acc = lowr ? qvc : 0.0, acc2 = 0.0;
#pragma unroll 8
        for (int j = 0; j < NP; j += 2) {
            acc = fma(-sA[j * SAS + ls], sh[j], acc);
            acc2 = fma(-sA[(j + 1) * SAS + ls], sh[j + 1], acc2);
        }
        xb = lowr ? acc + acc2 : 0.0;
    }

    // W~ leaves the registers for the Lemke phase: it takes Ad's place in the block buffer, [x row][constraint], for the read-back
    // (the post-check reads Ad from memory again: the records of this node are a few microseconds old in L2 / the memory-side cache)
    wsync();
#pragma unroll
    for (int I = 0; I < T3; ++I)
#pragma unroll
        for (int Jb = 0; Jb < T3; ++Jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) sA[(16 * I + 4 * g + lq) * SAS + 16 * Jb + lc] = T[I][T3 + Jb][g];

    STAMP(2);   // S product, c, W~ parked
    // ================= Stage B: Lemke on the Schur dictionary (48 pairs, tile layout) =================
    // pair k (k < 48) <-> item row n + k:  p_k = (S lambda + c)_k in [l_k, u_k],  d_k = lambda_k.
    // ids: p_k -> k, d_k -> 48 + k, artificial -> 96; column index 48 = the extra (covering) column.
    constexpr int NBP = NP, XC = NP, VTH = 2 * NP;
    const bool actb = l < NBP;
    if (qpn_ballot(actb && lo == hi)) { decline(); return; }      // equality rows need their multiplier crashed in: general kernel
    const double lo0 = lo, hi0 = hi;
    const int clsv = (lo0 == -QINF && hi0 == QINF) ? 2 : 0;
    const double rngv = hi0 - lo0;
    int satv = 0;
    int rowvar = actb ? l : -1, colvar = actb ? NBP + l : (l == XC ? VTH : -1);
    double nbval = 0.0, tcol = 0.0;
    wsync();
    // the dictionary: SJ<Jb>[4 Ib + g] = row 16 Ib + 4 g + lq, column 16 Jb + lc -- three 16-element register vectors (12 used):
    // static element accesses are plain registers, the pivot row is read with a wave-uniform DYNAMIC index (s_set_gpr_idx:
    // no branch tree, no scratch), and the exchange is one asm block per column tile, so that every dictionary register has ONE
    // definition per iteration (a C++ switch over named scalars compiled to 334 register copies per pivot)
    typedef double d16 __attribute__((ext_vector_type(16)));
    d16 SJ0, SJ1, SJ2;
#define SDV(J, K) SJ##J[K]
#define FOR_K(M, J) M(J, 0) M(J, 1) M(J, 2) M(J, 3) M(J, 4) M(J, 5) M(J, 6) M(J, 7) M(J, 8) M(J, 9) M(J, 10) M(J, 11)
#define FOR_JK(M) FOR_K(M, 0) FOR_K(M, 1) FOR_K(M, 2)
#define M_INIT(J, K) SDV(J, K) = SB[(K) >> 2][J][(K) & 3];
    FOR_JK(M_INIT)
#undef M_INIT
    SJ0[12] = SJ0[13] = SJ0[14] = SJ0[15] = 0.0; SJ1[12] = SJ1[13] = SJ1[14] = SJ1[15] = 0.0; SJ2[12] = SJ2[13] = SJ2[14] = SJ2[15] = 0.0;
    auto col_of = [&](int v) -> int { return wave_first(colvar == v); };
    int pivots = n;
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;
    int status = QPN_FAILURE;
    int c = XC;
    bool sneg = true;
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
    // row i of a column sits at (i & 3) * 12 + 4 * (i >> 4) + ((i >> 2) & 3): a lane group reads its 12 rows as 12 consecutive doubles
    const int myslot = (l & 3) * 12 + ((l >> 4) << 2) + ((l >> 2) & 3);
    while (status == QPN_MAX_ITERS) {
        if (pivots >= max_piv) break;
        c = uni(c);
        // ---- entering column -> sucol
        if (c == XC) { if (actb) sucol[myslot] = tcol; }
        else if (lc == (c & 15)) {
            double *const dst = sucol + lq * 12;
            const int Jc = c >> 4;
#define M_PUB(J, K) dst[K] = SDV(J, K);
            if (Jc == 0) { FOR_K(M_PUB, 0) } else if (Jc == 1) { FOR_K(M_PUB, 1) } else { FOR_K(M_PUB, 2) }
#undef M_PUB
        }
        wsync();
        const double cm = actb ? sucol[myslot] : 0.0;
        STAMP(6);   // (diagnostic builds: entering column through LDS)
        // ---- ratio test (two-pass Harris with 1e-10 slack; largest pivot among ties, the artificial first)
        const double gdir = __hiloint2double(__double2hiint(cm) ^ (sneg ? (int)0x80000000 : 0), __double2loint(cm));
        const double rc = rcp64_(gdir);
        const bool gneg = __double2hiint(gdir) < 0;
        const double tb = gneg ? lo : hi;
        const bool cnd = actb && (fabs(gdir) > ptol) && (fabs(tb) < QINF);
        const double arc = fabs(rc);
        const double dd = cnd ? (tb - xb) * rc : QINF;
        const double d1 = fma(slack, arc, dd);
        const double dmax = fmin(wave_min_f64(d1), self_lim);
        if (uni(__double2hiint(dmax)) == 0x7ff00000 && uni(__double2loint(dmax)) == 0) { status = QPN_RAY_TERM; break; }
        const unsigned long long bal = qpn_ballot(dd <= dmax);
        STAMP(7);   // (... ratio test)
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, inv = 0.0;
        int rsel = 0, rq = 0, cnext = -1;
        bool pivoted = false, stop = false;
        int rW = 63, veW = 0, cW = 63, vlW = 0, kW = 63, auW = 0;
        double eloW = 0.0, ehiW = 0.0, nbW = 0.0;
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
            rq = r & 3;
            rsel = ((r >> 4) << 2) | ((r >> 2) & 3);              // register 4 Ib + g of row r
            {
                const double p0 = SJ0[rsel], p1 = SJ1[rsel], p2 = SJ2[rsel];          // uniform dynamic register index
                if (lq == rq) { svrow[lc] = p0; svrow[16 + lc] = p1; svrow[32 + lc] = p2; }
            }
            // the row's extra-column entry rides along as "column 48"; the entry of the pivot column itself is replaced by -1 so
            // that row * inv carries -inv there
            if (l == r) { svrow[XC] = tcol; svrow[c] = -1.0; }      // (one lane, program order after the owners' stores above)
            double step = readlane_f64(dd, r);
            if (step < 0.0) step = 0.0;
            const double leave_val = readlane_f64(tb, r);
            const double rcr = readlane_f64(rc, r);
            inv = sneg ? -rcr : rcr;
            const double delta = sneg ? -step : step;
            const int vl = readlane_i32(rowvar, r);
            const double enter_val = readlane_f64(nbval, c) + delta;
            wsync();
            {
                const double pa = svrow[lc], pb = svrow[16 + lc], pc = svrow[32 + lc], px = svrow[XC];
                v0 = pa * inv; v1 = pb * inv; v2 = pc * inv;
                const double vx = px * inv;
                const double tc0 = (c == XC) ? 0.0 : tcol;
                double xbn = fma(delta, cm, xb);
                double tcn = fma(-cm, vx, tc0);
                if (l == r) { xbn = enter_val; tcn = -vx; }
                xb = xbn; tcol = tcn;
            }
            pivoted = true;
            rW = r; veW = readlane_i32(colvar, c); eloW = elo; ehiW = ehi;
            cW = c; vlW = vl; nbW = leave_val;
            pivots++;
            if (vl == VTH) { status = QPN_SUCCESS; stop = true; }
            else {
                int vn;
                const int k = vl < NBP ? vl : vl - NBP;
                const int cls = readlane_i32(clsv, k);
                int au = readlane_i32(satv, k);
                if (vl < NBP) {
                    au = uni(__double2hiint(rcr)) >= 0 ? 1 : 0; kW = k; auW = au;
                    vn = NBP + k;
                    sneg = au != 0;
                    self_lim = QINF;
                    if (cls == 2) { elo = 0.0; ehi = 0.0; }
                    else if (au) { elo = -QINF; ehi = 0.0; }
                    else { elo = 0.0; ehi = QINF; }
                } else {
                    vn = k;
                    sneg = au != 0;
                    self_lim = readlane_f64(rngv, k);
                    if (cls == 2) sneg = false;
                    elo = readlane_f64(lo0, k); ehi = readlane_f64(hi0, k);
                }
                cnext = (vn == veW) ? -1 : col_of(vn);
                if (cnext < 0) { status = QPN_FAILURE; stop = true; }
            }
        }
        // the exchange's column operands: requested here so that the LDS trip overlaps the write-backs below
        const d4 *const up = reinterpret_cast<const d4 *>(sucol + lq * 12);
        const d4 ua = up[0], ub = up[1], uc = up[2];
        __builtin_amdgcn_sched_barrier(0);
        // ---- write-back of the single-lane bookkeeping updates (a write that does not apply goes to idle lane 63)
        if (l == rW) { rowvar = veW; lo = eloW; hi = ehiW; }
        if (l == cW) colvar = vlW;
        if (l == c) nbval = nbW;
        if (l == kW) satv = auW;
        if (stop) break;
        // ---- the exchange: ONE asm block per column tile, run every iteration (a bound flip runs it with v = 0 and empty lane
        // masks: a no-op), so the dictionary registers are never copied, selected or spilled whatever the control flow around
        {
            const double inv_s = udbl(inv);
            const int cx = uni(pivoted ? c : XC), rs = uni(rsel);
            const unsigned long long mrow = pivoted ? 0xFFFFull << (16 * rq) : 0ull;
            const unsigned long long mcol = (pivoted && cx < XC) ? 0x0001000100010001ull << (cx & 15) : 0ull;
#define M_XBLOCK(J, VJ)                                                                                                             \
            asm volatile(                                                                                                           \
                "v_fma_f64 %[s0], -%[u0], %[v], %[s0]\n\tv_fma_f64 %[s1], -%[u1], %[v], %[s1]\n\t"                                  \
                "v_fma_f64 %[s2], -%[u2], %[v], %[s2]\n\tv_fma_f64 %[s3], -%[u3], %[v], %[s3]\n\t"                                  \
                "v_fma_f64 %[s4], -%[u4], %[v], %[s4]\n\tv_fma_f64 %[s5], -%[u5], %[v], %[s5]\n\t"                                  \
                "v_fma_f64 %[s6], -%[u6], %[v], %[s6]\n\tv_fma_f64 %[s7], -%[u7], %[v], %[s7]\n\t"                                  \
                "v_fma_f64 %[s8], -%[u8], %[v], %[s8]\n\tv_fma_f64 %[s9], -%[u9], %[v], %[s9]\n\t"                                  \
                "v_fma_f64 %[s10], -%[u10], %[v], %[s10]\n\tv_fma_f64 %[s11], -%[u11], %[v], %[s11]\n\t"                            \
                /* column c (this tile's iff c >> 4 == J): T[i][c] = u_i * inv on the 4 lanes that own it */                       \
                "s_cmp_eq_u32 %[ct], " #J "\n\ts_cbranch_scc0 20f\n\t"                                                              \
                "s_mov_b64 exec, %[mc]\n\t"                                                                                         \
                "v_mul_f64 %[s0], %[u0], %[iv]\n\tv_mul_f64 %[s1], %[u1], %[iv]\n\tv_mul_f64 %[s2], %[u2], %[iv]\n\t"             \
                "v_mul_f64 %[s3], %[u3], %[iv]\n\tv_mul_f64 %[s4], %[u4], %[iv]\n\tv_mul_f64 %[s5], %[u5], %[iv]\n\t"             \
                "v_mul_f64 %[s6], %[u6], %[iv]\n\tv_mul_f64 %[s7], %[u7], %[iv]\n\tv_mul_f64 %[s8], %[u8], %[iv]\n\t"             \
                "v_mul_f64 %[s9], %[u9], %[iv]\n\tv_mul_f64 %[s10], %[u10], %[iv]\n\tv_mul_f64 %[s11], %[u11], %[iv]\n"            \
                "20:\n\t"                                                                                                           \
                /* row r: T[r][j] = -v_j on the 16 lanes that own it (v carries -inv at column c) */                                \
                "s_mov_b64 exec, %[mr]\n\t"                                                                                         \
                "s_cmp_gt_i32 %[rs], 5\n\ts_cbranch_scc1 26f\n\t"                                                                  \
                "s_cmp_gt_i32 %[rs], 2\n\ts_cbranch_scc1 23f\n\t"                                                                  \
                "s_cmp_eq_u32 %[rs], 0\n\ts_cbranch_scc0 21f\n\tv_mul_f64 %[s0], %[v], -1.0\n\ts_branch 32f\n"                     \
                "21:\n\ts_cmp_eq_u32 %[rs], 1\n\ts_cbranch_scc0 22f\n\tv_mul_f64 %[s1], %[v], -1.0\n\ts_branch 32f\n"             \
                "22:\n\tv_mul_f64 %[s2], %[v], -1.0\n\ts_branch 32f\n"                                                             \
                "23:\n\ts_cmp_eq_u32 %[rs], 3\n\ts_cbranch_scc0 24f\n\tv_mul_f64 %[s3], %[v], -1.0\n\ts_branch 32f\n"             \
                "24:\n\ts_cmp_eq_u32 %[rs], 4\n\ts_cbranch_scc0 25f\n\tv_mul_f64 %[s4], %[v], -1.0\n\ts_branch 32f\n"             \
                "25:\n\tv_mul_f64 %[s5], %[v], -1.0\n\ts_branch 32f\n"                                                             \
                "26:\n\ts_cmp_gt_i32 %[rs], 8\n\ts_cbranch_scc1 29f\n\t"                                                          \
                "s_cmp_eq_u32 %[rs], 6\n\ts_cbranch_scc0 27f\n\tv_mul_f64 %[s6], %[v], -1.0\n\ts_branch 32f\n"                     \
                "27:\n\ts_cmp_eq_u32 %[rs], 7\n\ts_cbranch_scc0 28f\n\tv_mul_f64 %[s7], %[v], -1.0\n\ts_branch 32f\n"             \
                "28:\n\tv_mul_f64 %[s8], %[v], -1.0\n\ts_branch 32f\n"                                                             \
                "29:\n\ts_cmp_eq_u32 %[rs], 9\n\ts_cbranch_scc0 30f\n\tv_mul_f64 %[s9], %[v], -1.0\n\ts_branch 32f\n"             \
                "30:\n\ts_cmp_eq_u32 %[rs], 10\n\ts_cbranch_scc0 31f\n\tv_mul_f64 %[s10], %[v], -1.0\n\ts_branch 32f\n"           \
                "31:\n\tv_mul_f64 %[s11], %[v], -1.0\n"                                                                            \
                "32:\n\ts_mov_b64 exec, -1"                                                                                        \
                : [s0] "+v"(SDV(J, 0)), [s1] "+v"(SDV(J, 1)), [s2] "+v"(SDV(J, 2)), [s3] "+v"(SDV(J, 3)), [s4] "+v"(SDV(J, 4)),    \
                  [s5] "+v"(SDV(J, 5)), [s6] "+v"(SDV(J, 6)), [s7] "+v"(SDV(J, 7)), [s8] "+v"(SDV(J, 8)), [s9] "+v"(SDV(J, 9)),    \
                  [s10] "+v"(SDV(J, 10)), [s11] "+v"(SDV(J, 11))                                                                  \
                : [u0] "v"(ua[0]), [u1] "v"(ua[1]), [u2] "v"(ua[2]), [u3] "v"(ua[3]), [u4] "v"(ub[0]), [u5] "v"(ub[1]),            \
                  [u6] "v"(ub[2]), [u7] "v"(ub[3]), [u8] "v"(uc[0]), [u9] "v"(uc[1]), [u10] "v"(uc[2]), [u11] "v"(uc[3]),          \
                  [v] "v"(VJ), [iv] "s"(inv_s), [mc] "s"(mcol), [mr] "s"(mrow), [ct] "s"(cx >> 4), [rs] "s"(rs)                    \
                : "scc");
            M_XBLOCK(0, v0) M_XBLOCK(1, v1) M_XBLOCK(2, v2)
#undef M_XBLOCK
        }
        c = cnext;
        wsync();
        STAMP(3);   // (... row, bookkeeping, exchange)
    }

    STAMP(3);   // Lemke
    // ---- read back: lambda_k, then x = W~ lambda - h -------------------------------------------------------------
    wsync();
    if (actb) sval[rowvar] = xb;
    if (l <= XC) sval[colvar] = nbval;
    wsync();
    // x_l = (W~ lambda)_l - h_l, lane l <-> row l: W~ from the block buffer, lambda broadcast from sval
    double wl = 0.0;
    {
        double w2 = 0.0;
        const int ls = l < NP ? l : 0;
#pragma unroll 8
        for (int k = 0; k < NP; k += 2) {
            wl = fma(sA[ls * SAS + k], sval[NBP + k], wl);
            w2 = fma(sA[ls * SAS + k + 1], sval[NBP + k + 1], w2);
        }
        wl += w2;
    }
    // item order: rows < n are x, rows n.. are lambda (two rounds of 64 item rows)
    double zk0 = 0.0, zk1 = 0.0;
    {
        const int i0 = l, i1 = 64 + l;
        if (i0 < N) zk0 = i0 < n ? wl - kx : sval[NBP + (i0 - n)];
        if (i1 < N) zk1 = sval[NBP + (i1 - n)];                  // (n <= 48 < 64: the second round holds multipliers only)
    }
    wsync();
    if (l < N) SZ(l) = zk0;
    if (64 + l < N) SZ(64 + l) = zk1;
    wsync();

    STAMP(4);   // read-back
    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -------------------------------
    int bad = 0;
    double nres = 0.0;
    const double tol = a.check_tol, ct = a.comp_tol;
    auto check_row = [&](int k, double zk, double qk) {
        const bool isx = k < n;
        double rk = qk;
        // a row's terms in ascending order on two alternating accumulators, 24 loads in flight together: the blocks are in L2 / the
        // memory-side cache and a row is a chain of round trips, not bandwidth (8 in flight were 18 trips per node; 24 are 6:
        // + 0.5 / 1.5 / 2.7 % on the class at 33 / 40 / 48)
#define CHK_DOT(PB, PTR, STRIDE, CNT, ZOFF, ACC_E, ACC_O, SGN)                                                                      \
        for (; j + PB <= (CNT); j += PB) {                                                                                          \
            double mv[PB];                                                                                                          \
            _Pragma("unroll") for (int q8 = 0; q8 < PB; ++q8) mv[q8] = (PTR)[(size_t)(j + q8) * (STRIDE)];                          \
            _Pragma("unroll") for (int q8 = 0; q8 < PB; q8 += 2) {                                                                  \
                ACC_E = fma(SGN mv[q8], SZ((ZOFF) + j + q8), ACC_E); ACC_O = fma(SGN mv[q8 + 1], SZ((ZOFF) + j + q8 + 1), ACC_O);    \
            }                                                                                                                       \
        }
        double r2 = 0.0;
        if (isx) {
            const double *qcol = Q_ + k;
            int j = 0;
            CHK_DOT(24, qcol, n, n, 0, rk, r2, +)
            CHK_DOT(8, qcol, n, n, 0, rk, r2, +)
            for (; j < n; ++j) rk = fma(qcol[(size_t)j * n], SZ(j), rk);
            const double *arow = A_ + (size_t)k * m;                                      // column k of Ad: A[i][k], i contiguous; columns of lambda: -A'
            j = 0;
            CHK_DOT(24, arow, 1, m, n, r2, rk, -)
            CHK_DOT(8, arow, 1, m, n, r2, rk, -)
            for (; j < m; ++j) r2 = fma(-arow[j], SZ(n + j), r2);
        } else {
            const double *acol = A_ + (k - n);                                            // row r of Ad: A[r][j], lanes <-> r contiguous
            int j = 0;
            CHK_DOT(24, acol, m, n, 0, rk, r2, +)
            CHK_DOT(8, acol, m, n, 0, rk, r2, +)
            for (; j < n; ++j) rk = fma(acol[(size_t)j * m], SZ(j), rk);
        }
        rk += r2;
#undef CHK_DOT
        const int gk = !isx;
        const double lk = gk ? a.nd.l[(size_t)b * m + (k - n)] : -QINF, uk = gk ? a.nd.u[(size_t)b * m + (k - n)] : QINF;
        const double pp = gk ? rk : zk, dv = gk ? zk : rk;
        if (dv > tol && fabs(pp - lk) > tol) bad++;
        if (dv < -tol && fabs(pp - uk) > tol) bad++;
        if (pp - lk < -tol) bad++;
        if (pp - uk > tol) bad++;
        if (isnan(pp) || isnan(dv)) bad++;
        double tt = pp - dv;
        if (tt < lk) tt = lk;
        if (tt > uk) tt = uk;
        double e = fabs(pp - tt);
        if (isnan(e)) e = QINF;
        nres = fmax(nres, e);
        unsigned mask = 0;
        auto approx = [&](double x, double y) { return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= ct); };
        if (!approx(lk, uk)) {
            if (approx(pp, lk) && dv >= -ct) mask |= 1u;
            if (lk - ct <= pp && pp <= uk + ct && fabs(dv) <= ct) mask |= 2u;
            if (approx(pp, uk) && dv <= ct) mask |= 4u;
        } else mask = 8u;
        if (gk) mask <<= 4;
        a.z[(size_t)b * N + k] = zk;
        if (a.x && isx) {
            const size_t xo = (size_t)b * (size_t)a.stride_x + k;
            a.x[xo] = zk;
            for (int q = 0; q < a.n_mirror; ++q) a.mirror[q][xo] = zk;
        }
        if (a.active) a.active[(size_t)b * N + k] = (uint8_t)mask;
    };
    if (l < N) check_row(l, zk0, qv0);
    if (64 + l < N) check_row(64 + l, zk1, qv1);
    bad = wave_sum_i32(bad);
    nres = wave_max_f64(nres);
    if (bad > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
    if (l == 0) {
        a.status[b] = status;
        if (a.resid) a.resid[b] = nres;
        if (a.pivots) a.pivots[b] = pivots;
        if (a.sched_key) { const int k0 = a.sched_key[b]; a.sched_key[b] = k0 > 0 ? k0 - (k0 >> 5) + pivots : 32 * pivots; }
    }
    STAMP(5);   // post-check + stores
#ifdef QPN_STAMPS
    if (a.stamps && l == 0) for (int i = 0; i < 8; ++i) a.stamps[(size_t)b * 8 + i] = stamp_acc[i];
#endif
#undef SZ
#undef SDV
#undef FOR_K
#undef FOR_JK
}

} // namespace

bool qpn_schur48_shape(int n, int m) { return n >= 1 && m >= 0 && n <= 48 && m <= 48 && (n > 32 || m > 32); }

hipError_t qpn_launch_avi_solve_schur48_nodes(const AviBatchArgs &a, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(avi_solve_schur48, dim3((unsigned)a.batch), dim3(64), 0, stream, a);
    return hipGetLastError();
}
