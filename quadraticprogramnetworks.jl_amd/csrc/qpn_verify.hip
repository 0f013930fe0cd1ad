// qpn_verify.hip -- batched per-node KKT verification for gfx950 (CDNA4).
//
// Restates verify_solution, src/qp_processing.jl:57-149, on dense per-node records
// (Qd, R, qd, Ad, B, l, u) with the reference's tolerances: feasibility 1e-3 (:86), active
// rows 1e-2 (:98-99), sign / residual `tol` = 1e-4 (:57, :119), fallback residual 1e-4 (:138).
//
// Three launches on one stream, no host round trip:
//   verify_stage1   one wavefront per node: gradient q~ (:58-60), Ax (:84), feasibility,
//                   active-row classes, least-squares duals by the normal equations
//                   G y = A_bar' q~ with a diagonally pivoted Cholesky kept in LDS (the sparse
//                   `\` of :115), sign/residual test (:119).  Nodes that fail it get their
//                   bounded-LSQ box-AVI (Ad Ad', -Ad q~, sign bounds; :129-137 with the PATH MCP
//                   of :12-27 reduced to its lambda block) written to scratch, path = -1.
//   avi_solve_lds1  (qpn_avi_solve.hip) gated on path == -1.
//   verify_stage2   accepts iff || Ad' lambda - q~ ||_2 <= 1e-4 (:138).
#include "qpn_internal.h"
#include "qpn_tile_chol.h"
#include <type_traits>

#define QINF __builtin_huge_val()

namespace {

constexpr int WAVE = 64;
// Leading dimension of the two LDS matrices (odd: column and row sweeps are both conflict-light).  Two size
// classes: n, m <= 32 -> 33 (17 KB of LDS per node, 9 nodes resident per CU); n, m <= 64 -> 65 (67 KB, 2 per CU).

struct VerifyArgs {
    int32_t batch, n, m, p;
    const double *Qd, *R, *qd, *Ad, *B, *l, *u, *xd, *w;
    int64_t stride_w;
    double tol;
    int32_t *solution;
    double *lambda;
    int32_t *path;
    double *sG, *sq, *slb, *sub, *sz;
    int32_t gate;        // verify_stage1 only: != 0 -> take just the nodes whose path is -2 (left over by verify_node64)
};

// Pivot choice of every factorisation below: the largest remaining diagonal, where diagonals within 2^-30 of the largest count as
// equal and the lowest column wins.  With equilibrated rows every diagonal starts at 1 up to rounding; an exact `==` would let
// that rounding pick the pivot, and the basic solution of a rank-deficient block (which multipliers are 0) would depend on it.
// (The CPU restatement used by the tests applies the same rule.)
constexpr double PIV_BAND = 1.0 - 0x1p-30;

template <int LDV>
__global__ __launch_bounds__(WAVE) void verify_stage1(VerifyArgs a)
{
    const int n = a.n, m = a.m, p = a.p;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (a.gate && a.path[b] != -2) return;
    __shared__ double sA[LDV * (LDV - 1)];   // Ad, column-major, ld = LDV
    __shared__ double sGa[LDV * (LDV - 1)];  // Gram block of the active rows, then its Cholesky factor
    __shared__ double sqt[64];        // q~
    __shared__ double svec[64];       // broadcast vector (xd, factor column, y)
    __shared__ int srow[64];          // active column -> row
    __shared__ double ssg[64];        // active column -> sign

    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *B_ = a.B + (size_t)b * m * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double *lam = a.lambda + (size_t)b * m;

    if (lane < n) svec[lane] = a.xd[(size_t)b * n + lane];
    if (lane < m)
        for (int j = 0; j < n; ++j) sA[j * LDV + lane] = A_[(size_t)j * m + lane];
    __syncthreads();

    // :58-60  q~ = Q[dec,:] x + q[dec]   (lane i < n)
    double qt = 0.0;
    if (lane < n) {
        qt = a.qd[(size_t)b * n + lane];
        for (int j = 0; j < n; ++j) qt = fma(Q_[(size_t)j * n + lane], svec[j], qt);
        for (int k = 0; k < p; ++k) qt = fma(R_[(size_t)k * n + lane], w_[k], qt);
        sqt[lane] = qt;
    }
    // :84  ax (lane r < m)
    double ax = 0.0, lr = 0.0, ur = 0.0;
    if (lane < m) {
        for (int j = 0; j < n; ++j) ax = fma(sA[j * LDV + lane], svec[j], ax);
        for (int k = 0; k < p; ++k) ax = fma(B_[(size_t)k * m + lane], w_[k], ax);
        lr = a.l[(size_t)b * m + lane]; ur = a.u[(size_t)b * m + lane];
        lam[lane] = 0.0;
    }
    __syncthreads();

    // :86  feasibility, tol 1e-3 (Slice membership, src/sets.jl:851-854)
    const bool infeas = lane < m && !(lr - 1e-3 <= ax && ax - 1e-3 <= ur);
    if (__ballot(infeas)) {
        if (lane == 0) { a.solution[b] = 0; a.path[b] = 0; }
        return;
    }
    if (m == 0) {   // :91-96
        double s = wave_sum_f64(lane < n ? qt * qt : 0.0);
        if (lane == 0) { a.solution[b] = sqrt(s) <= a.tol ? 1 : 0; a.path[b] = 1; }
        return;
    }
    // :98-103  active-row classes
    const bool pos0 = lane < m && ax < lr + 1e-2, neg0 = lane < m && ax > ur - 1e-2;
    const int cls = (pos0 ? 1 : 0) | (neg0 ? 2 : 0);
    const unsigned long long bp = __ballot(cls == 1), bn = __ballot(cls == 2), bb = __ballot(cls == 3);
    const int np = __popcll(bp), nn = __popcll(bn), nb = __popcll(bb);
    const int k = np + nn + nb;
    const unsigned long long below = (1ull << lane) - 1ull;
    int mycol = -1;
    if (cls == 1) mycol = __popcll(bp & below);
    else if (cls == 2) mycol = np + __popcll(bn & below);
    else if (cls == 3) mycol = np + nn + __popcll(bb & below);
    if (mycol >= 0) { srow[mycol] = lane; ssg[mycol] = cls == 2 ? -1.0 : 1.0; }
    __syncthreads();

    // ---- :114-115  least squares  A_bar y ~ q~  through the normal equations --------------
    // lane c < k owns active column c:  A_bar(:,c) = sg_c * Ad(row_c,:)'
    const int myrow = lane < k ? srow[lane] : 0;
    const double mysg = lane < k ? ssg[lane] : 0.0;
    double rhs = 0.0;
    // the k x k Gram entries are spread over all 64 lanes (k lanes with k entries each would leave most of the wave
    // idle); every entry is the same ascending fma chain over t as before
    for (int e = lane; e < k * k; e += WAVE) {
        const int c1 = e % k, c2 = e / k;
        const int r1 = srow[c1], r2 = srow[c2];
        double s = 0.0;
        for (int t = 0; t < n; ++t) s = fma(sA[t * LDV + r1], sA[t * LDV + r2], s);
        sGa[c2 * LDV + c1] = s * ssg[c1] * ssg[c2];
    }
    if (lane < k) {
        for (int t = 0; t < n; ++t) rhs = fma(sA[t * LDV + myrow], sqt[t], rhs);
        rhs *= mysg;
    }
    __syncthreads();
    // diagonally pivoted Cholesky, in place: column ord[s] of sGa becomes column s of the factor
    bool done = !(lane < k);       // this lane's column already pivoted (or not a column)
    int mystep = -1;               // elimination step at which this lane was the pivot
    double diag = lane < k ? sGa[lane * LDV + lane] : 0.0;
    const double dscale = wave_max_f64(lane < k ? diag : 0.0);
    int rank = 0;
    double bvec = rhs;             // running right-hand side of the forward substitution
    for (int s = 0; s < k; ++s) {
        const double dtop = wave_max_f64(done ? -1.0 : diag);
        if (!(dtop > 1e-12 * (dscale > 1.0 ? dscale : 1.0))) break;
        const int pv = wave_first(!done && diag >= dtop * PIV_BAND);
        const double dmax = __shfl(diag, pv, WAVE);
        const double lpp = sqrt(dmax);
        // factor column: L(i,s) = G(i,pv)/lpp for the remaining i; L(pv,s) = lpp
        double lis = 0.0;
        if (lane < k && !done) lis = lane == pv ? lpp : sGa[pv * LDV + lane] / lpp;
        if (lane < k) svec[lane] = (done || lane == pv) ? 0.0 : lis;
        __syncthreads();
        if (lane < k && !done && lane != pv) {
            for (int j = 0; j < k; ++j) {
                const double lj = svec[j];
                if (lj != 0.0) sGa[j * LDV + lane] = fma(-lis, lj, sGa[j * LDV + lane]);
            }
            diag = sGa[lane * LDV + lane];
        }
        if (lane < k && !done) sGa[pv * LDV + lane] = lis;   // store L(:,s) in the freed column
        // forward substitution step: w_s = b_pv / lpp ; b_i -= L(i,s) w_s
        const double ws = __shfl(bvec, pv, WAVE) / lpp;
        if (lane < k && !done && lane != pv) bvec = fma(-lis, ws, bvec);
        if (lane == pv) { done = true; mystep = s; bvec = ws; }
        rank++;
        __syncthreads();
    }
    // back substitution  L' y = w  over the pivoted columns, last step first
    double y = 0.0;
    for (int s = rank - 1; s >= 0; --s) {
        const int pv = wave_first(mystep == s);
        // sum over columns pivoted later: L(i, s) * y_i, with L(:,s) stored in column pv
        const double term = (lane < k && mystep > s) ? sGa[pv * LDV + lane] * y : 0.0;
        const double acc = wave_sum_f64(term);
        if (lane == pv) y = (bvec - acc) / sGa[pv * LDV + pv];
    }
    // :119  sign and residual tests
    const bool badsign = lane < np + nn && !(y > -a.tol);
    __syncthreads();
    if (lane < k) svec[lane] = y * mysg;
    __syncthreads();
    double res = 0.0;
    if (lane < n) {
        double s = -qt;
        for (int c = 0; c < k; ++c) s = fma(sA[lane * LDV + srow[c]], svec[c], s);
        res = s * s;
    }
    res = wave_sum_f64(res);
    const bool ok = !__ballot(badsign) && sqrt(res) <= a.tol;
    if (ok) {
        if (lane < k) lam[myrow] = (mysg < 0.0) ? -y : y;   // :120-123
        if (lane == 0) { a.solution[b] = 1; a.path[b] = 2; }
        return;
    }
    // ---- :129-137  bounded least squares as a box-AVI in lambda, handed to the AVI kernel ---
    if (lane < m) {
        double *G = a.sG + (size_t)b * m * m;
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int t = 0; t < n; ++t) s = fma(sA[t * LDV + lane], sA[t * LDV + j], s);
            G[(size_t)j * m + lane] = s;
        }
        double s = 0.0;
        for (int t = 0; t < n; ++t) s = fma(sA[t * LDV + lane], sqt[t], s);
        a.sq[(size_t)b * m + lane] = -s;
        a.slb[(size_t)b * m + lane] = (cls & 2) ? -QINF : 0.0;   // :129-131
        a.sub[(size_t)b * m + lane] = (cls & 1) ? QINF : 0.0;    // :132-134
        a.sz[(size_t)b * m + lane] = 0.0;
    }
    if (lane == 0) { a.solution[b] = 0; a.path[b] = -1; }
}

// ---- verify_node32: the n, m <= 32 class rebuilt (round 4) ---------------------------------------------------------------
// One wavefront per node, ALL 64 lanes at work, 9.9 KB of LDS (four waves per SIMD), the Gram block on the matrix cores:
//   * Qd and Ad arrive with fully coalesced loads, two columns of 32 rows per instruction (512 contiguous bytes at n = m = 32),
//     all 32 + the vector loads in flight at once; they STAY in registers (lane (r, h) holds row r, columns of parity h);
//   * q~ = Qd x + R w + qd and ax = Ad x + B w are two 16-term chains per lane over both halves of the wavefront, x read from
//     LDS de-interleaved by parity (8 ds_read_b128), the halves added with one v_permlane32_swap pair;
//   * the signed active rows go to LDS compacted (row c of A_bar', leading dimension 34: conflict-free as MFMA operands),
//     G = A_bar' A_bar is 8 (k <= 16) or 24 v_mfma_f64_16x16x4_f64 -- an operand register is A AND B operand of a diagonal
//     tile -- and lands in the same LDS region (leading dimension 33), which the factor then takes over column by column;
//   * the diagonally pivoted Cholesky is LEFT-looking: step s forms column pv of the Schur complement from the factor's
//     earlier columns (2 s independent LDS reads, one chain of s fma) instead of updating k columns of G through LDS
//     round trips; same pivots, same arithmetic per entry as the right-looking form of verify_stage1;
//   * the back substitution is column-oriented (no wave reduction per step), the residual reads the re-staged active rows.
// Decisions (feasibility, classes, pivot order, sign and residual tests) are verify_stage1's; q~, ax and the Gram entries differ
// from its ascending chains by summation order only.
typedef double vd4 __attribute__((ext_vector_type(4)));
typedef double vd2 __attribute__((ext_vector_type(2)));
#define VMFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)

__device__ __forceinline__ void vwave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double vsum_halves(double v)      // v[l] + v[l ^ 32] in every lane
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
__device__ __forceinline__ double vadd2(double a, double b) { return a + b; }
__device__ __forceinline__ double wave_sum32_f64(double v)   // sum over lanes 0..31 (lanes 32..63 ignored), wave-uniform
{
    QPN_ROW_REDUCE(v, vadd2);
    return readlane_f64(v, 0) + readlane_f64(v, 16);
}

constexpr int V32_LDA = 34;      // compacted active rows / all rows: entry (row c, column t) at c * 34 + t
constexpr int V32_LDG = 33;      // Gram block / factor: entry (i, j) at j * 33 + i

template <bool FULL>
__global__ __launch_bounds__(WAVE, 4) void verify_node32(VerifyArgs a)
{
    const int n = FULL ? 32 : a.n, m = FULL ? 32 : a.m, p = a.p;
    const int l = threadIdx.x;
    const int b = blockIdx.x;
    const int r5 = l & 31, ch = l >> 5;
    __shared__ __attribute__((aligned(16))) double sM[32 * V32_LDA];
    __shared__ __attribute__((aligned(16))) double sx[32];       // x, then q~: [parity][16]
    __shared__ __attribute__((aligned(16))) double sv[32];       // broadcast vector (rhs, y)
    __shared__ double sd[32];          // 1 / |row| of the active rows, by column

    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *B_ = a.B + (size_t)b * m * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double *lam = a.lambda + (size_t)b * m;

    // ---- every load of the node in flight at once
    double vq[16], va[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int cj = 2 * t + ch;
        const bool okq = FULL || (cj < n && r5 < n), oka = FULL || (cj < n && r5 < m);
        vq[t] = Q_[okq ? (size_t)cj * n + r5 : 0];
        va[t] = A_[oka ? (size_t)cj * m + r5 : 0];
    }
    const double xv = (l < n) ? a.xd[(size_t)b * n + l] : 0.0;
    const double qdv = (FULL || r5 < n) ? a.qd[(size_t)b * n + r5] : 0.0;
    double lr = 0.0, ur = 0.0;
    if (FULL || r5 < m) { lr = a.l[(size_t)b * m + r5]; ur = a.u[(size_t)b * m + r5]; }
    double accr = 0.0, accb = 0.0;                            // (R w)_r and (B w)_r: columns of parity ch
    for (int k0 = ch; k0 < p; k0 += 8) {
        double rv[4], bv[4], wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kk = k0 + 2 * k;
            const bool ok = kk < p;
            wv[k] = ok ? w_[kk] : 0.0;
            rv[k] = (ok && (FULL || r5 < n)) ? R_[(size_t)kk * n + r5] : 0.0;
            bv[k] = (ok && (FULL || r5 < m)) ? B_[(size_t)kk * m + r5] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { accr = fma(rv[k], wv[k], accr); accb = fma(bv[k], wv[k], accb); }
    }
    if (l < 32) sx[(l & 1) * 16 + (l >> 1)] = xv;
    vwave_sync();
    if (!FULL) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int cj = 2 * t + ch;
            if (!(cj < n && r5 < n)) vq[t] = 0.0;
            if (!(cj < n && r5 < m)) va[t] = 0.0;
        }
    }
    // ---- :58-60 q~, :84 ax: 16 terms per lane, both halves of the wavefront
    double q0 = accr, q1 = 0.0, a0 = accb, a1 = 0.0;
    {
        const vd2 *xh = reinterpret_cast<const vd2 *>(sx + ch * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const vd2 xx = xh[j];
            q0 = fma(vq[2 * j], xx[0], q0); q1 = fma(vq[2 * j + 1], xx[1], q1);
            a0 = fma(va[2 * j], xx[0], a0); a1 = fma(va[2 * j + 1], xx[1], a1);
        }
    }
    const double qt = vsum_halves(q0 + q1) + qdv;             // q~ of row r5, in both halves
    const double ax = vsum_halves(a0 + a1);                   // (Ad x + B w) of row r5, in both halves
    const bool isrow = FULL || r5 < m;

    // :86  feasibility, tol 1e-3 (Slice membership, src/sets.jl:851-854)
    const bool infeas = isrow && !(lr - 1e-3 <= ax && ax - 1e-3 <= ur);
    if (qpn_ballot(infeas)) {
        if (l < m) lam[l] = 0.0;
        if (l == 0) { a.solution[b] = 0; a.path[b] = 0; }
        return;
    }
    if (!FULL && m == 0) {   // :91-96
        const double s = wave_sum32_f64((l < n) ? qt * qt : 0.0);
        if (l == 0) { a.solution[b] = sqrt(s) <= a.tol ? 1 : 0; a.path[b] = 1; }
        return;
    }
    // :98-103  active-row classes (same in both halves)
    const bool pos0 = isrow && ax < lr + 1e-2, neg0 = isrow && ax > ur - 1e-2;
    const int cls = (pos0 ? 1 : 0) | (neg0 ? 2 : 0);
    const unsigned bp = (unsigned)qpn_ballot(cls == 1), bn = (unsigned)qpn_ballot(cls == 2), bb = (unsigned)qpn_ballot(cls == 3);
    const int np = __popc(bp), nn = __popc(bn), nb = __popc(bb);
    const int k = np + nn + nb;
    const unsigned below = (1u << r5) - 1u;
    int mycol = -1;
    if (cls == 1) mycol = __popc(bp & below);
    else if (cls == 2) mycol = np + __popc(bn & below);
    else if (cls == 3) mycol = np + nn + __popc(bb & below);
    const double mysgr = (cls == 2) ? -1.0 : 1.0;             // sign of row r5 as a column of A_bar
    // The active rows enter the least-squares problem EQUILIBRATED: row r scaled to unit length, y'_c = |row| y_c.  A solution
    // graph's rows are normalised to a leading coefficient of 1 (src/sets.jl:76-89), so a piece row with a small leading entry is
    // 1e7 times longer than its neighbours; unscaled, the Gram block's diagonal spans 1e14 and the factorisation's rank test
    // (relative to the LARGEST diagonal) drops every ordinary column.  Signs, ranks and the residual A_bar y - q~ do not depend on
    // the scaling; y itself is scaled back for the sign test's tolerance and for lambda.
    double rn0 = 0.0, rn1 = 0.0;
#pragma unroll
    for (int t = 0; t < 16; t += 2) { rn0 = fma(va[t], va[t], rn0); rn1 = fma(va[t + 1], va[t + 1], rn1); }
    const double rn2 = vsum_halves(rn0 + rn1);
    const double dinv = rn2 > 0.0 ? 1.0 / sqrt(rn2) : 0.0;
    const double rsc = mysgr * dinv;
    // ---- A_bar' (signed, scaled active rows, compacted) and q~ to LDS
    vwave_sync();                                             // (the x reads are done)
    if (mycol >= 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) sM[mycol * V32_LDA + 2 * t + ch] = rsc * va[t];
        if (ch == 0) sd[mycol] = dinv;
    }
    if (l < 32) sx[(l & 1) * 16 + (l >> 1)] = (FULL || l < n) ? qt : 0.0;
    vwave_sync();
    // (Ad q~)_r for every row (the fallback's -Ad q~ too), then the right-hand side A_bar' q~ by column
    double g0 = 0.0, g1 = 0.0;
    {
        const vd2 *qh = reinterpret_cast<const vd2 *>(sx + ch * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const vd2 qq = qh[j];
            g0 = fma(va[2 * j], qq[0], g0); g1 = fma(va[2 * j + 1], qq[1], g1);
        }
    }
    const double aq = vsum_halves(g0 + g1);                   // (Ad q~)_r5
    if (mycol >= 0 && ch == 0) sv[mycol] = rsc * aq;
    vwave_sync();
    const double rhs = (l < k) ? sv[l] : 0.0;                 // A_bar' q~ by column (lane c < k <-> column c)
    const double dcol = (l < k) ? sd[l] : 0.0;                // this column's scale
    const int lc = l & 15, lq = l >> 4;
    const bool two = k > 16;                                  // (wave-uniform)
    const bool mine = l < k;

    // Least squares  A_bar(:, P) y ~ q~  for a column set P (lane c: inP): the Gram block on the matrix cores (operand lane
    // (lc, lq) <-> A_bar'(row lc [+16], column 4 s + lq), read from the staged active rows), then the diagonally pivoted
    // Cholesky, left-looking, on the same LDS region, forward and back substitution.  Returns y (0 outside P and on the
    // columns the factorisation drops as dependent) and whether this lane's column was pivoted.  On entry the region holds
    // the active rows (leading dimension 34); on exit the factor.
    auto lsq_on = [&](bool inP, double &y_out, bool &pivoted) {
        double x0[8], x1[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int t = 4 * s + lq;
            const double v0 = sM[lc * V32_LDA + t];
            x0[s] = (lc < k && (FULL || t < n)) ? v0 : 0.0;
        }
        if (two) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int t = 4 * s + lq;
                const double v1 = sM[(16 + lc) * V32_LDA + t];
                x1[s] = (16 + lc < k && (FULL || t < n)) ? v1 : 0.0;
            }
        }
        vwave_sync();                                         // operands are in; the region changes hands
        {
            vd4 g00 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 8; ++s) g00 = VMFMA(x0[s], x0[s], g00);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = lq + 4 * g, j = lc;
                if (i < k && j < k) sM[j * V32_LDG + i] = g00[g];
            }
            if (two) {
                vd4 g01 = {0.0, 0.0, 0.0, 0.0}, g11 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 8; ++s) { g01 = VMFMA(x0[s], x1[s], g01); g11 = VMFMA(x1[s], x1[s], g11); }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = lq + 4 * g, j = 16 + lc;
                    if (j < k) { sM[j * V32_LDG + i] = g01[g]; sM[i * V32_LDG + j] = g01[g]; }
                    if (16 + i < k && j < k) sM[j * V32_LDG + 16 + i] = g11[g];
                }
            }
        }
        vwave_sync();
        // lane i < k <-> row i.  Column ord[s] of the region becomes column s of the factor (the freed column), as in verify_stage1.
        bool done = !inP;
        int mystep = -1;
        double diag = inP ? sM[l * V32_LDG + l] : 0.0;
        const double dscale = wave_max_f64(inP ? diag : 0.0);
        const double dfloor = 1e-12 * (dscale > 1.0 ? dscale : 1.0);
        int rank = 0;
        int ordv = 0;                                         // lane s holds the pivot column of step s
        double bvec = inP ? rhs : 0.0, myinv = 0.0;
        for (int s = 0; s < k; ++s) {
            const double dtop = wave_max_f64(done ? -1.0 : diag);
            if (!(dtop > dfloor)) break;
            const int pv = wave_first(!done && diag >= dtop * PIV_BAND);
            const double dmax = __shfl(diag, pv, WAVE);
            const double lpp = sqrt(dmax), inv = 1.0 / lpp;
            double acc = done ? 0.0 : sM[pv * V32_LDG + l];   // G(i, pv)
            for (int t0 = 0; t0 < s; t0 += 4) {
                double av[4], bw[4];
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) {
                    const int t = t0 + u_;
                    const int o = readlane_i32(ordv, t < s ? t : s - 1);
                    av[u_] = sM[o * V32_LDG + (mine ? l : 0)];
                    bw[u_] = (t < s) ? sM[o * V32_LDG + pv] : 0.0;
                }
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) acc = fma(-av[u_], bw[u_], acc);
            }
            const double lis = (l == pv) ? lpp : acc * inv;
            if (!done) sM[pv * V32_LDG + l] = lis;            // L(:, s) into the freed column
            const double ws = readlane_f64(bvec, pv) * inv;   // forward substitution: w_s = b_pv / lpp ; b_i -= L(i, s) w_s
            if (!done && l != pv) { diag = fma(-lis, lis, diag); bvec = fma(-lis, ws, bvec); }
            if (l == pv) { done = true; mystep = s; bvec = ws; myinv = inv; }
            if (l == s) ordv = pv;
            rank++;
            vwave_sync();
        }
        // back substitution L' y = w, column-oriented: y of the last pivot first; a lane pivoted at step a < s' takes
        // L(ord[s'], a) -- row ord[s'] of its own (freed) column -- times y_{ord[s']} off its right-hand side
        double y = 0.0;
        for (int s = rank - 1; s >= 0; --s) {
            const int pvs = readlane_i32(ordv, s);
            const double ys = readlane_f64(bvec * myinv, pvs);
            if (mystep >= 0 && mystep < s) bvec = fma(-sM[l * V32_LDG + pvs], ys, bvec);
            if (l == pvs) y = ys;
        }
        y_out = y; pivoted = mystep >= 0;
        vwave_sync();                                         // (the factor's reads are done)
    };
    auto stage_active_rows = [&]() {
        if (mycol >= 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) sM[mycol * V32_LDA + 2 * t + ch] = rsc * va[t];
        }
    };
    // r = A_bar y - q~ by lane t < n (y handed over through sv), on the staged active rows; returns this lane's entry
    auto residual_entry = [&](double y) -> double {
        if (l < k) sv[l] = y;
        vwave_sync();
        double s0 = (l < 32) ? -qt : 0.0;
        const int tcol = (l < 32 && (FULL || l < n)) ? l : 0;
        for (int c0 = 0; c0 < k; c0 += 4) {
            double av[4], yv[4];
#pragma unroll
            for (int u_ = 0; u_ < 4; ++u_) {
                const int c = c0 + u_;
                av[u_] = sM[(c < k ? c : k - 1) * V32_LDA + tcol];
                yv[u_] = (c < k) ? sv[c] : 0.0;
            }
#pragma unroll
            for (int u_ = 0; u_ < 4; ++u_) s0 = fma(av[u_], yv[u_], s0);
        }
        return (l < 32 && (FULL || l < n)) ? s0 : 0.0;
    };

    // ---- :114-115  least squares on ALL active columns
    double y; bool piv;
    lsq_on(mine, y, piv);
    // :119  sign test; then the residual A_bar y - q~ on the re-staged active rows
    const bool badsign = l < np + nn && !(y * dcol > -a.tol);
    stage_active_rows();
    double rt = residual_entry(y);
    {
        const double res = wave_sum32_f64(rt * rt);
        const bool ok = !qpn_ballot(badsign) && sqrt(res) <= a.tol;
        if (ok) {
            if (l < m) lam[l] = (mycol >= 0) ? rsc * sv[mycol] : 0.0;      // :120-123 (scaled back)
            if (l == 0) { a.solution[b] = 1; a.path[b] = 2; }
            return;
        }
    }
    // ---- :129-137  bounded least squares  min |Ad' lambda - q~|  with the sign bounds of :129-134, in the signed columns:
    // y_c >= 0 on the pos / neg columns (c < np + nn), y_c free on the equality columns, lambda = 0 on the inactive rows.  The
    // reference hands this QP to PATH (solve_qp, :12-33); its minimiser's image Ad' lambda is unique, so the accept test of
    // :138 does not depend on the method.  Here: Lawson-Hanson's active-set iteration started from the full column set --
    // the least-squares solution just computed -- every re-solve a fresh Gram block on the matrix cores and the same
    // Cholesky: the node never leaves the wavefront (no scratch block, no second and third launch).
    {
        const bool cons = l < np + nn;                        // sign-constrained column
        bool inP = mine && piv;                               // passive set: the columns the factorisation kept
        bool blocked = false;                                 // a column that came back non-positive after entering (LH's guard)
        double ycur = 0.0, sl = y;                            // feasible iterate (starts at 0), least-squares solution on P
        int iters = 0;
        const int cap = 3 * k + 12;
        bool failed = false;
        for (;;) {
            // inner loop: bring the least-squares solution on P into the feasible set
            for (;;) {
                const bool bad = inP && cons && !(sl > 0.0);
                if (!qpn_ballot(bad)) { ycur = inP ? sl : 0.0; break; }
                const double ratio = bad ? ycur / (ycur - sl) : QINF;          // ycur >= 0 >= sl: in [0, 1]; 0/0 -> NaN -> leaves
                const double alpha = wave_min_f64((bad && ratio == ratio) ? ratio : (bad ? 0.0 : QINF));
                if (inP) ycur = fma(alpha, sl - ycur, ycur);
                const bool leave = bad && (!(ratio == ratio) || ratio <= alpha);      // the blocking column(s) go to zero
                if (leave) { inP = false; ycur = 0.0; }
                if (++iters > cap) { failed = true; break; }
                stage_active_rows();
                vwave_sync();
                bool pv2;
                lsq_on(inP, sl, pv2);
                if (inP && !pv2) { inP = false; ycur = 0.0; }                       // dependent on the others: out, at zero
            }
            if (failed) break;
            // outer test: the gradient  w = A_bar' (q~ - A_bar y)  on the zero set
            stage_active_rows();
            rt = residual_entry(ycur);
            if (l < 32) sx[l] = rt;                           // (sx is free: q~ lives in registers by now)
            vwave_sync();
            double wgr = 0.0;
            if (mine) {
                const vd2 *rowc = reinterpret_cast<const vd2 *>(sM + l * V32_LDA);
                const vd2 *rr = reinterpret_cast<const vd2 *>(sx);
                double w0 = 0.0, w1 = 0.0;
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) {
                    const vd2 aa = rowc[t2], r2 = rr[t2];
                    w0 = fma(aa[0], r2[0], w0); w1 = fma(aa[1], r2[1], w1);
                }
                wgr = -(w0 + w1);                             // A_bar_c . (q~ - A_bar y)
            }
            const double gscale = wave_max_f64(mine ? fabs(rhs) : 0.0);
            const bool cand = mine && cons && !inP && !blocked && wgr > 1e-11 * (gscale > 1.0 ? gscale : 1.0);
            if (!qpn_ballot(cand)) break;                     // optimal
            const double wmax = wave_max_f64(cand ? wgr : -1.0);
            const int enter = wave_first(cand && wgr == wmax);
            if (l == enter) inP = true;
            if (++iters > cap) { failed = true; break; }
            vwave_sync();
            bool pv2;
            lsq_on(inP, sl, pv2);
            if (inP && !pv2) { inP = false; if (l == enter) blocked = true; }
            const double s_enter = readlane_f64(sl, enter);
            if (!(s_enter > 0.0)) {                           // numerically it should be: leave it out for good and carry on
                if (l == enter) { inP = false; blocked = true; }
                stage_active_rows();
                vwave_sync();
                lsq_on(inP, sl, pv2);
                if (inP && !pv2) inP = false;
            }
        }
        if (failed) {                                         // :143-145
            if (l < m) lam[l] = 0.0;
            if (l == 0) { a.solution[b] = 0; a.path[b] = 5; }
            return;
        }
        // :138  accept iff | Ad' lambda - q~ |_2 <= 1e-4  (rt is the residual at ycur, from the last outer test)
        const double res = wave_sum32_f64(rt * rt);
        const bool ok = sqrt(res) <= 1e-4;
        if (l < k) sv[l] = ycur;
        vwave_sync();
        if (l < m) lam[l] = (mycol >= 0) ? rsc * sv[mycol] : 0.0;
        if (l == 0) { a.solution[b] = ok ? 1 : 0; a.path[b] = ok ? 3 : 4; }
    }
}

// ---- verify_node64: the same design for n, m <= 64 (one of them > 32), one wavefront per node, lane r <-> row r ------------
// Qd and Ad stream through registers in panels of 16 columns (whole 512-byte columns, coalesced): pass 1 forms q~ and ax, pass 2
// (Ad again: L2 / Infinity Cache) stages the signed active rows in LDS and forms Ad q~.  Up to 32 active rows (nodes with more are
// flagged -2 and taken by verify_stage1<65> in a gated launch behind this one): the Gram block is 16 or 48 MFMAs over 16
// k-steps, the factor and the bounded least-squares fallback are verify_node32's.  The active rows keep their own LDS region
// (16.9 KB) next to the Gram block / factor (8.4 KB): 26.6 KB per wavefront, six wavefronts per CU.
constexpr int V64_LDA = 66;      // active rows: entry (row c, column t) at c * 66 + t (operand reads conflict-free)

__global__ __launch_bounds__(WAVE, 2) void verify_node64(VerifyArgs a)
{
    const int n = a.n, m = a.m, p = a.p;
    const int l = threadIdx.x;
    const int b = blockIdx.x;
    __shared__ __attribute__((aligned(16))) double sM[32 * V64_LDA];
    __shared__ __attribute__((aligned(16))) double sG[32 * V32_LDG];
    __shared__ __attribute__((aligned(16))) double sx[64];       // x, then q~, then the residual
    __shared__ __attribute__((aligned(16))) double sv[32];
    __shared__ double sd[32];          // 1 / |row| of the active rows, by column (see verify_node32: the rows are equilibrated)

    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *B_ = a.B + (size_t)b * m * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double *lam = a.lambda + (size_t)b * m;
    const bool isx = l < n, isrow = l < m;
    const int nch = (n + 15) >> 4;

    sx[l] = isx ? a.xd[(size_t)b * n + l] : 0.0;
    double qt = isx ? a.qd[(size_t)b * n + l] : 0.0;
    double ax = 0.0, lr = 0.0, ur = 0.0;
    if (isrow) { lr = a.l[(size_t)b * m + l]; ur = a.u[(size_t)b * m + l]; }
    for (int k0 = 0; k0 < p; k0 += 4) {
        double rv[4], bv[4], wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kk = k0 + k;
            const bool ok = kk < p;
            wv[k] = ok ? w_[kk] : 0.0;
            rv[k] = (ok && isx) ? R_[(size_t)kk * n + l] : 0.0;
            bv[k] = (ok && isrow) ? B_[(size_t)kk * m + l] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { qt = fma(rv[k], wv[k], qt); ax = fma(bv[k], wv[k], ax); }
    }
    vwave_sync();
    // ---- pass 1: q~ (:58-60) and ax (:84), 16 columns of Qd and of Ad in flight per panel
    // (the rows' squared lengths come out of the same values: a second pass over Ad for them and for Ad q~ is not needed -- the
    //  right-hand side is only wanted on the active rows, which pass 2 reads anyway)
    double rn2 = 0.0;
    {
        double q1 = 0.0, a1 = 0.0, r1 = 0.0;
        for (int c = 0; c < nch; ++c) {
            double vq[16], va[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int cj = 16 * c + t;
                vq[t] = (cj < n && isx) ? Q_[(size_t)cj * n + l] : 0.0;
                va[t] = (cj < n && isrow) ? A_[(size_t)cj * m + l] : 0.0;
            }
            const vd2 *xh = reinterpret_cast<const vd2 *>(sx + 16 * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const vd2 xx = xh[j];
                qt = fma(vq[2 * j], xx[0], qt); q1 = fma(vq[2 * j + 1], xx[1], q1);
                ax = fma(va[2 * j], xx[0], ax); a1 = fma(va[2 * j + 1], xx[1], a1);
                rn2 = fma(va[2 * j], va[2 * j], rn2); r1 = fma(va[2 * j + 1], va[2 * j + 1], r1);
            }
        }
        qt += q1; ax += a1; rn2 += r1;
    }
    // :86  feasibility, tol 1e-3
    const bool infeas = isrow && !(lr - 1e-3 <= ax && ax - 1e-3 <= ur);
    if (qpn_ballot(infeas)) {
        if (isrow) lam[l] = 0.0;
        if (l == 0) { a.solution[b] = 0; a.path[b] = 0; }
        return;
    }
    // :98-103  active-row classes
    const bool pos0 = isrow && ax < lr + 1e-2, neg0 = isrow && ax > ur - 1e-2;
    const int cls = (pos0 ? 1 : 0) | (neg0 ? 2 : 0);
    const unsigned long long bp = qpn_ballot(cls == 1), bn = qpn_ballot(cls == 2), bb = qpn_ballot(cls == 3);
    const int np = __popcll(bp), nn = __popcll(bn), nb = __popcll(bb);
    const int k = np + nn + nb;
    if (k > 32) {                                             // more active rows than this kernel's Gram block holds
        if (l == 0) { a.solution[b] = 0; a.path[b] = -2; }
        return;
    }
    const unsigned long long below = (1ull << l) - 1ull;
    int mycol = -1;
    if (cls == 1) mycol = __popcll(bp & below);
    else if (cls == 2) mycol = np + __popcll(bn & below);
    else if (cls == 3) mycol = np + nn + __popcll(bb & below);
    const double mysgr = (cls == 2) ? -1.0 : 1.0;
    vwave_sync();                                             // (the x reads are done)
    sx[l] = isx ? qt : 0.0;
    vwave_sync();
    // ---- pass 2: the active rows of Ad again (L2 / Infinity Cache): (Ad q~)_r, and the signed, equilibrated rows to LDS
    double aq = 0.0;
    const double dinv = rn2 > 0.0 ? 1.0 / sqrt(rn2) : 0.0;
    const double rsc = mysgr * dinv;
    if (qpn_ballot(mycol >= 0)) {
        for (int c = 0; c < nch; ++c) {
            double va[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int cj = 16 * c + t;
                va[t] = (cj < n && mycol >= 0) ? A_[(size_t)cj * m + l] : 0.0;
            }
            if (mycol >= 0) {
#pragma unroll
                for (int t = 0; t < 16; ++t) sM[mycol * V64_LDA + 16 * c + t] = rsc * va[t];
                const vd2 *qh = reinterpret_cast<const vd2 *>(sx + 16 * c);
                double g1 = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const vd2 qq = qh[j]; aq = fma(va[2 * j], qq[0], aq); g1 = fma(va[2 * j + 1], qq[1], g1); }
                aq += g1;
            }
        }
    }
    if (mycol >= 0) { sv[mycol] = rsc * aq; sd[mycol] = dinv; }
    vwave_sync();
    const double rhs = (l < k) ? sv[l] : 0.0;
    const double dcol = (l < k) ? sd[l] : 0.0;
    const int lc = l & 15, lq = l >> 4;
    const bool two = k > 16;
    const bool mine = l < k;
    const int nks = (n + 3) >> 2;                             // k-steps of the Gram block (wave-uniform)

    auto lsq_on = [&](bool inP, double &y_out, bool &pivoted) {
        {
            vd4 g00 = {0.0, 0.0, 0.0, 0.0}, g01 = {0.0, 0.0, 0.0, 0.0}, g11 = {0.0, 0.0, 0.0, 0.0};
            for (int s0 = 0; s0 < nks; s0 += 4) {
                double x0[4], x1[4];
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) {
                    const int t = 4 * (s0 + u_) + lq;
                    const bool okt = s0 + u_ < nks && t < n;
                    const double v0 = sM[lc * V64_LDA + (okt ? t : 0)];
                    x0[u_] = (lc < k && okt) ? v0 : 0.0;
                    const double v1 = two ? sM[(16 + lc) * V64_LDA + (okt ? t : 0)] : 0.0;
                    x1[u_] = (two && 16 + lc < k && okt) ? v1 : 0.0;
                }
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) {
                    g00 = VMFMA(x0[u_], x0[u_], g00);
                    if (two) { g01 = VMFMA(x0[u_], x1[u_], g01); g11 = VMFMA(x1[u_], x1[u_], g11); }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = lq + 4 * g, j = lc;
                if (i < k && j < k) sG[j * V32_LDG + i] = g00[g];
                if (two) {
                    const int j2 = 16 + lc;
                    if (j2 < k) { sG[j2 * V32_LDG + i] = g01[g]; sG[i * V32_LDG + j2] = g01[g]; }
                    if (16 + i < k && j2 < k) sG[j2 * V32_LDG + 16 + i] = g11[g];
                }
            }
        }
        vwave_sync();
        bool done = !inP;
        int mystep = -1;
        double diag = inP ? sG[l * V32_LDG + l] : 0.0;
        const double dscale = wave_max_f64(inP ? diag : 0.0);
        const double dfloor = 1e-12 * (dscale > 1.0 ? dscale : 1.0);
        int rank = 0;
        int ordv = 0;
        double bvec = inP ? rhs : 0.0, myinv = 0.0;
        for (int s = 0; s < k; ++s) {
            const double dtop = wave_max_f64(done ? -1.0 : diag);
            if (!(dtop > dfloor)) break;
            const int pv = wave_first(!done && diag >= dtop * PIV_BAND);
            const double dmax = __shfl(diag, pv, WAVE);
            const double lpp = sqrt(dmax), inv = 1.0 / lpp;
            double acc = done ? 0.0 : sG[pv * V32_LDG + l];
            for (int t0 = 0; t0 < s; t0 += 4) {
                double av[4], bw[4];
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) {
                    const int t = t0 + u_;
                    const int o = readlane_i32(ordv, t < s ? t : s - 1);
                    av[u_] = sG[o * V32_LDG + (mine ? l : 0)];
                    bw[u_] = (t < s) ? sG[o * V32_LDG + pv] : 0.0;
                }
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) acc = fma(-av[u_], bw[u_], acc);
            }
            const double lis = (l == pv) ? lpp : acc * inv;
            if (!done) sG[pv * V32_LDG + l] = lis;
            const double ws = readlane_f64(bvec, pv) * inv;
            if (!done && l != pv) { diag = fma(-lis, lis, diag); bvec = fma(-lis, ws, bvec); }
            if (l == pv) { done = true; mystep = s; bvec = ws; myinv = inv; }
            if (l == s) ordv = pv;
            rank++;
            vwave_sync();
        }
        double y = 0.0;
        for (int s = rank - 1; s >= 0; --s) {
            const int pvs = readlane_i32(ordv, s);
            const double ys = readlane_f64(bvec * myinv, pvs);
            if (mystep >= 0 && mystep < s) bvec = fma(-sG[l * V32_LDG + pvs], ys, bvec);
            if (l == pvs) y = ys;
        }
        y_out = y; pivoted = mystep >= 0;
        vwave_sync();
    };
    // r = A_bar y - q~, lane t < n
    auto residual_entry = [&](double y) -> double {
        if (l < k) sv[l] = y;
        vwave_sync();
        double s0 = -qt;
        const int tcol = isx ? l : 0;
        for (int c0 = 0; c0 < k; c0 += 4) {
            double av[4], yv[4];
#pragma unroll
            for (int u_ = 0; u_ < 4; ++u_) {
                const int c = c0 + u_;
                av[u_] = sM[(c < k ? c : k - 1) * V64_LDA + tcol];
                yv[u_] = (c < k) ? sv[c] : 0.0;
            }
#pragma unroll
            for (int u_ = 0; u_ < 4; ++u_) s0 = fma(av[u_], yv[u_], s0);
        }
        return isx ? s0 : 0.0;
    };

    double y; bool piv;
    lsq_on(mine, y, piv);
    const bool badsign = l < np + nn && !(y * dcol > -a.tol);
    double rt = residual_entry(y);
    {
        const double res = wave_sum_f64(rt * rt);
        const bool ok = !qpn_ballot(badsign) && sqrt(res) <= a.tol;
        if (ok) {
            if (isrow) lam[l] = (mycol >= 0) ? rsc * sv[mycol] : 0.0;
            if (l == 0) { a.solution[b] = 1; a.path[b] = 2; }
            return;
        }
    }
    // ---- :129-137  bounded least squares inside the wavefront (verify_node32's iteration)
    {
        const bool cons = l < np + nn;
        bool inP = mine && piv;
        bool blocked = false;
        double ycur = 0.0, sl = y;
        int iters = 0;
        const int cap = 3 * k + 12;
        bool failed = false;
        for (;;) {
            for (;;) {
                const bool bad = inP && cons && !(sl > 0.0);
                if (!qpn_ballot(bad)) { ycur = inP ? sl : 0.0; break; }
                const double ratio = bad ? ycur / (ycur - sl) : QINF;
                const double alpha = wave_min_f64((bad && ratio == ratio) ? ratio : (bad ? 0.0 : QINF));
                if (inP) ycur = fma(alpha, sl - ycur, ycur);
                const bool leave = bad && (!(ratio == ratio) || ratio <= alpha);
                if (leave) { inP = false; ycur = 0.0; }
                if (++iters > cap) { failed = true; break; }
                bool pv2;
                lsq_on(inP, sl, pv2);
                if (inP && !pv2) { inP = false; ycur = 0.0; }
            }
            if (failed) break;
            rt = residual_entry(ycur);
            vwave_sync();
            sx[l] = rt;                                       // (q~ lives in registers by now)
            vwave_sync();
            double wgr = 0.0;
            if (mine) {
                const vd2 *rowc = reinterpret_cast<const vd2 *>(sM + l * V64_LDA);
                const vd2 *rr = reinterpret_cast<const vd2 *>(sx);
                double w0 = 0.0, w1 = 0.0;
                for (int t2 = 0; t2 < (n + 1) / 2; ++t2) {
                    const vd2 aa = rowc[t2], r2 = rr[t2];
                    w0 = fma(aa[0], r2[0], w0);
                    if (2 * t2 + 1 < n) w1 = fma(aa[1], r2[1], w1);
                }
                wgr = -(w0 + w1);
            }
            const double gscale = wave_max_f64(mine ? fabs(rhs) : 0.0);
            const bool cand = mine && cons && !inP && !blocked && wgr > 1e-11 * (gscale > 1.0 ? gscale : 1.0);
            if (!qpn_ballot(cand)) break;
            const double wmax = wave_max_f64(cand ? wgr : -1.0);
            const int enter = wave_first(cand && wgr == wmax);
            if (l == enter) inP = true;
            if (++iters > cap) { failed = true; break; }
            bool pv2;
            lsq_on(inP, sl, pv2);
            if (inP && !pv2) { inP = false; if (l == enter) blocked = true; }
            const double s_enter = readlane_f64(sl, enter);
            if (!(s_enter > 0.0)) {
                if (l == enter) { inP = false; blocked = true; }
                lsq_on(inP, sl, pv2);
                if (inP && !pv2) inP = false;
            }
        }
        if (failed) {
            if (isrow) lam[l] = 0.0;
            if (l == 0) { a.solution[b] = 0; a.path[b] = 5; }
            return;
        }
        const double res = wave_sum_f64(rt * rt);
        const bool ok = sqrt(res) <= 1e-4;
        if (l < k) sv[l] = ycur;
        vwave_sync();
        if (isrow) lam[l] = (mycol >= 0) ? rsc * sv[mycol] : 0.0;
        if (l == 0) { a.solution[b] = ok ? 1 : 0; a.path[b] = ok ? 3 : 4; }
    }
}

__global__ __launch_bounds__(WAVE) void verify_stage2(VerifyArgs a, const int32_t *avi_status)
{
    const int n = a.n, m = a.m, p = a.p;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (a.path[b] != -1) return;
    double *lam = a.lambda + (size_t)b * m;
    if (avi_status[b] != QPN_SUCCESS) {   // :143-145
        if (lane < m) lam[lane] = 0.0;
        if (lane == 0) { a.solution[b] = 0; a.path[b] = 5; }
        return;
    }
    __shared__ double sl_[64], sx[64];
    if (lane < m) { sl_[lane] = a.sz[(size_t)b * m + lane]; lam[lane] = sl_[lane]; }
    if (lane < n) sx[lane] = a.xd[(size_t)b * n + lane];
    __syncthreads();
    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double res = 0.0;
    if (lane < n) {
        double qt = a.qd[(size_t)b * n + lane];
        for (int j = 0; j < n; ++j) qt = fma(Q_[(size_t)j * n + lane], sx[j], qt);
        for (int k = 0; k < p; ++k) qt = fma(R_[(size_t)k * n + lane], w_[k], qt);
        double s = -qt;
        for (int i = 0; i < m; ++i) s = fma(A_[(size_t)lane * m + i], sl_[i], s);
        res = s * s;
    }
    res = wave_sum_f64(res);
    if (lane == 0) {
        const bool ok = sqrt(res) <= 1e-4;   // :138
        a.solution[b] = ok ? 1 : 0;
        a.path[b] = ok ? 3 : 4;
    }
}

// ---- wide nodes (n or m beyond 64, up to 512): one workgroup of 256 threads per node -----------------------------
// The same steps as verify_stage1 with the matrices left in global memory (Ad of a 256 x 256 node is 512 KB): thread c
// owns the active columns c, c + 256 of the Gram matrix, which lives in the node's m x m scratch block (sG, leading
// dimension m) and is factored there by the same diagonally pivoted Cholesky; reductions go over the workgroup.
constexpr int WTPB = 256, WMAX = 512;

__device__ __forceinline__ double block_sum_f64(double v, double *red)       // all threads get the sum; 2 barriers
{
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max_f64(double v, double *red)
{
    v = wave_max_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ---- verify_wide_node (round 4): n or m beyond 64 (up to 512), up to 128 active rows -- BASELINE config 5's nodes ----------------
// One workgroup of 256 threads per node.  Round 1's wide kernel (below: still the route of nodes with more than 128 active rows,
// gated) formed the Gram block entry by entry from global memory (two strided columns per entry) and factored it right-looking
// in global memory: 4.6 ms per 512 nodes of 256 x 256, 20 ms when every node takes the bounded least-squares fallback (a box-AVI of
// size m on the large-item kernel).  Here:
//   pass 1   q~ = Qd x + R w + qd and ax = Ad x + B w: thread <-> row (two rows per thread beyond 256), whole columns coalesced, four
//            accumulators per row; feasibility, classes, the compacted column order [pos | neg | both];
//   pass 2   Ad again (L2 / Infinity Cache), in panels of 16 (m <= 256) or 8 columns through LDS, double-buffered: the Gram block
//            of the active rows on the matrix cores -- wave v owns the tiles e = v (mod 4) of the upper triangle, at most nine,
//            in registers across the panels; an operand lane (lc, lq) reads panel column 4 s + lq at row srow[16 I + lc] -- and the
//            right-hand side A_bar' q~ by column from the same panels;
//   scaling  the rows are equilibrated through the Gram block's own diagonal (verify_node32), G goes to the node's workspace;
//   factor   diagonally pivoted Cholesky, left-looking, thread <-> row, L in a second workspace block so that G survives for the
//            fallback's re-solves; column-oriented back substitution;
//   tests    signs, residual A_bar y - q~ (thread <-> entry, the active rows' entries read from the records), lambda;
//   fallback verify_node32's active-set iteration, a block at a time: re-solves re-factor G's principal submatrix (no Gram pass).
constexpr int VW_KMAX = 128;
__device__ __forceinline__ int vw_ld(int m) { return m | 1; }

__global__ __launch_bounds__(WTPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void verify_wide_node(VerifyArgs a, double *gws, int wp, int dyn_doubles, int *slot_ctr, int nslots)
{
    const int n = a.n, m = a.m, p = a.p;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    __shared__ double sx[WMAX];                  // x, then q~, then the residual
    __shared__ double sdv[VW_KMAX], ssgv[VW_KMAX], sco[VW_KMAX];
    __shared__ int srow[VW_KMAX], sord[VW_KMAX];
    __shared__ unsigned char scls[WMAX];
    __shared__ double red[4];
    __shared__ int s_flag, s_k, s_np, s_nn, s_pv;
    __shared__ int s_cnt[3][8];
    __shared__ double s_bc[3];
    extern __shared__ __attribute__((aligned(16))) double dynp[];         // two panels [wp][ldp]
    const int ldp = vw_ld(m);

    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *B_ = a.B + (size_t)b * m * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double *lam = a.lambda + (size_t)b * m;
    const int mp = (m + 15) & ~15;
    // (behind verify_node64 -- a.gate set --: only the nodes it flagged, each with a slot of the small workspace; a node that finds no
    //  slot keeps its flag for round 1's kernels)
    int slot = b;
    if (a.gate) {
        if (a.path[b] != -2) return;
        if (tid == 0) s_flag = atomicAdd(slot_ctr, 1);
        __syncthreads();
        slot = s_flag;
        __syncthreads();
        if (slot >= nslots) return;
    }
    double *G = gws + (size_t)slot * 2 * mp * mp;       // k x k, leading dimension ldg
    double *L = G + (size_t)mp * mp;

    for (int i = tid; i < n; i += WTPB) sx[i] = a.xd[(size_t)b * n + i];
    if (tid == 0) s_flag = 0;
    __syncthreads();
    // ---- pass 1: q~ (:58-60), ax (:84)
    double qt[2] = {0.0, 0.0}, axv[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = tid + WTPB * h;
        if (i < n) {
            double s0 = a.qd[(size_t)b * n + i], s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = 0;
            for (; j + 16 <= n; j += 16) {                       // sixteen columns in flight
                double v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = Q_[(size_t)(j + t) * n + i];
#pragma unroll
                for (int t = 0; t < 16; t += 4) {
                    s0 = fma(v[t], sx[j + t], s0); s1 = fma(v[t + 1], sx[j + t + 1], s1);
                    s2 = fma(v[t + 2], sx[j + t + 2], s2); s3 = fma(v[t + 3], sx[j + t + 3], s3);
                }
            }
            for (; j < n; ++j) s0 = fma(Q_[(size_t)j * n + i], sx[j], s0);
            for (int k = 0; k < p; ++k) s1 = fma(R_[(size_t)k * n + i], w_[k], s1);
            qt[h] = (s0 + s1) + (s2 + s3);
        }
        if (i < m) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = 0;
            for (; j + 16 <= n; j += 16) {
                double v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = A_[(size_t)(j + t) * m + i];
#pragma unroll
                for (int t = 0; t < 16; t += 4) {
                    s0 = fma(v[t], sx[j + t], s0); s1 = fma(v[t + 1], sx[j + t + 1], s1);
                    s2 = fma(v[t + 2], sx[j + t + 2], s2); s3 = fma(v[t + 3], sx[j + t + 3], s3);
                }
            }
            for (; j < n; ++j) s0 = fma(A_[(size_t)j * m + i], sx[j], s0);
            for (int k = 0; k < p; ++k) s1 = fma(B_[(size_t)k * m + i], w_[k], s1);
            axv[h] = (s0 + s1) + (s2 + s3);
            const double lr = a.l[(size_t)b * m + i], ur = a.u[(size_t)b * m + i];
            if (!(lr - 1e-3 <= axv[h] && axv[h] - 1e-3 <= ur)) s_flag = 1;          // :86
            scls[i] = (unsigned char)(((axv[h] < lr + 1e-2) ? 1 : 0) | ((axv[h] > ur - 1e-2) ? 2 : 0));   // :98-103
        }
    }
    __syncthreads();
    if (s_flag) {
        for (int r = tid; r < m; r += WTPB) lam[r] = 0.0;
        if (tid == 0) { a.solution[b] = 0; a.path[b] = 0; }
        return;
    }
    // q~ to LDS (x is done with), active columns [pos | neg | both], rows ascending within a class
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int i = tid + WTPB * h; if (i < n) sx[i] = qt[h]; }
    // column of every active row: classes in the order [pos | neg | both] (:114), rows ascending within a class -- ballots per
    // wavefront and half (rows tid and tid + 256), offsets from a 3 x 8 table of counts
    {
        int mycls[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) { const int i = tid + WTPB * h; mycls[h] = i < m ? scls[i] : 0; }
        unsigned long long bal[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bal[c][h] = qpn_ballot(mycls[h] == c + 1);
                if (lane == 0) s_cnt[c][4 * h + wave] = __popcll(bal[c][h]);
            }
        __syncthreads();
        int tot[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { tot[c] = 0; for (int q = 0; q < 8; ++q) tot[c] += s_cnt[c][q]; }
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = mycls[h] - 1;
            if (c >= 0) {
                int pos = c == 0 ? 0 : (c == 1 ? tot[0] : tot[0] + tot[1]);
                for (int q = 0; q < 4 * h + wave; ++q) pos += s_cnt[c][q];
                pos += __popcll(bal[c][h] & below);
                if (pos < VW_KMAX) { srow[pos] = tid + WTPB * h; ssgv[pos] = c == 1 ? -1.0 : 1.0; }
            }
        }
        if (tid == 0) { s_k = tot[0] + tot[1] + tot[2]; s_np = tot[0]; s_nn = tot[1]; }
    }
    __syncthreads();
    const int k = s_k, npn = s_np + s_nn;
    if (k > VW_KMAX) {                                           // more active rows than the tiles of this kernel hold: round 1's route
        if (tid == 0) { a.solution[b] = 0; a.path[b] = -2; }
        return;
    }
    const int T = (k + 15) >> 4, ldg = 16 * T;
    // ---- pass 2: Gram block (raw) on the matrix cores, right-hand side by column, panels of wp columns
    vd4 acc[9];
    int tI[9], tJ[9];
    {
        int e = 0, cnt = 0;
        for (int I = 0; I < T; ++I)
            for (int J = I; J < T; ++J, ++e)
                if ((e & 3) == wave && cnt < 9) { tI[cnt] = I; tJ[cnt] = J; ++cnt; }
        for (; cnt < 9; ++cnt) { tI[cnt] = -1; tJ[cnt] = -1; }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = vd4{0.0, 0.0, 0.0, 0.0};
    double rhs = 0.0;
    const int npanel = (n + wp - 1) / wp;
    const int myrow = tid < k ? srow[tid] : 0;
    // (software-pipelined: the next panel's 16 values per thread are requested before the current panel's work and stored to the
    //  other buffer after it -- a load straight into LDS would stall the thread at the store until the data is there)
    double pre[16];
    auto fetch_panel = [&](int pn) {
        const int j0 = pn * wp;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = tid + WTPB * q;
            const int jj = idx / m, r = idx - jj * m;
            pre[q] = (idx < wp * m && j0 + jj < n) ? A_[(size_t)(j0 + jj) * m + r] : 0.0;
        }
    };
    auto store_panel = [&](double *buf) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = tid + WTPB * q;
            const int jj = idx / m, r = idx - jj * m;
            if (idx < wp * m) buf[jj * ldp + r] = pre[q];
        }
    };
    fetch_panel(0);
    store_panel(dynp);
    __syncthreads();
    for (int pn = 0; pn < npanel; ++pn) {
        double *cur = dynp + (size_t)(pn & 1) * wp * ldp, *nxt = dynp + (size_t)((pn + 1) & 1) * wp * ldp;
        if (pn + 1 < npanel) fetch_panel(pn + 1);
        const int j0 = pn * wp;
        if (tid < k) {
            double r1 = 0.0;
            for (int jj = 0; jj < wp; jj += 2) {
                rhs = fma(cur[jj * ldp + myrow], (j0 + jj < n) ? sx[j0 + jj] : 0.0, rhs);
                r1 = fma(cur[(jj + 1) * ldp + myrow], (j0 + jj + 1 < n) ? sx[j0 + jj + 1] : 0.0, r1);
            }
            rhs += r1;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (tI[q] < 0) continue;                             // (wave-uniform)
            const int ci = 16 * tI[q] + lc, cj = 16 * tJ[q] + lc;
            const int ri = ci < k ? srow[ci] : 0, rj = cj < k ? srow[cj] : 0;
            for (int s4 = 0; s4 < wp / 4; ++s4) {
                const double av = ci < k ? cur[(4 * s4 + lq) * ldp + ri] : 0.0;
                const double bv = cj < k ? cur[(4 * s4 + lq) * ldp + rj] : 0.0;
                acc[q] = VMFMA(av, bv, acc[q]);
            }
        }
        if (pn + 1 < npanel) store_panel(nxt);
        __syncthreads();
    }
    // (full-rank least squares first try the blocked, unpivoted factorisation of qpn_tile_chol.h on LDS tiles: k <= 112 -> 28 tiles
    //  = 61 KB of the dynamic block; a pivot below 1e-9 -- dependent active rows -- hands over to the pivoted factorisation below)
    const bool fast_chol = k <= 112 && tc_tiles(T) * TC_TSZ <= dyn_doubles;
    // ---- equilibration from the raw diagonal, then G (signed, scaled) to the workspace, both triangles
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        if (tI[q] < 0 || tI[q] != tJ[q]) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = 16 * tI[q] + lq + 4 * g, j = 16 * tJ[q] + lc;
            if (i == j && i < k) sdv[i] = acc[q][g] > 0.0 ? 1.0 / sqrt(acc[q][g]) : 0.0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        if (tI[q] < 0) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = 16 * tI[q] + lq + 4 * g, j = 16 * tJ[q] + lc;
            double v = 0.0;
            if (i < k && j < k) {
                v = acc[q][g] * (sdv[i] * ssgv[i]) * (sdv[j] * ssgv[j]);
                G[(size_t)j * ldg + i] = v;
                if (tI[q] != tJ[q]) G[(size_t)i * ldg + j] = v;
            }
            // ... and into the packed lower tiles of the blocked factorisation (the panels are done with): the accumulator holds
            // (row i of tile row tI, column j of tile column tJ >= tI), the lower tile (tJ, tI) takes its transpose; the padding up
            // to whole tiles is the identity
            if (fast_chol) dynp[tc_toff(tJ[q], tI[q]) + lc * TC_TLD + (lq + 4 * g)] = (i < k && j < k) ? v : (i == j ? 1.0 : 0.0);
        }
    }
    const bool mine = tid < k;
    const double dcol = mine ? sdv[tid] : 0.0;
    rhs = mine ? rhs * dcol * ssgv[tid] : 0.0;
    __threadfence_block();
    __syncthreads();

    // least squares on the column set P (thread c: inP): diagonally pivoted Cholesky of G(P, P), left-looking, L in its own block
    // (the factor lives in the dynamic LDS block -- the panels are done with -- when k x k doubles fit it: config 5's nodes have
    //  70 .. 95 active rows; otherwise in the workspace.  Two instantiations, so that each knows its address space.)
    const bool lds_factor = (long long)k * (k | 1) <= (long long)dyn_doubles;
    const int ldl = lds_factor ? (k | 1) : ldg;                   // (odd: a row of the factor, read across lanes, hits distinct banks)
    auto lsq_impl = [&](auto in_lds, bool inP, double &y_out, bool &pivoted) {
        double *Lf;
        if constexpr (decltype(in_lds)::value) Lf = dynp; else Lf = L;
        bool done = !inP;
        int mystep = -1;
        double diag = inP ? G[(size_t)tid * ldg + tid] : 0.0;
        double dsc = wave_max_f64(inP ? diag : 0.0);
        if (lane == 0) red[wave] = dsc;
        __syncthreads();
        const double dscale = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        const double dfloor = 1e-12 * (dscale > 1.0 ? dscale : 1.0);
        __syncthreads();
        int rank = 0;
        double bvec = inP ? rhs : 0.0, myinv = 0.0;
        for (int s = 0; s < k; ++s) {
            // pivot (PIV_BAND above): the remaining diagonals go to LDS (`sx` is free here), ONE barrier, then every wave works
            // the same pivot out of the k <= 128 values, two per lane
            if (tid < VW_KMAX) sx[tid] = (mine && !done) ? diag : -1.0;
            __syncthreads();                                      // A: (also: everything step s - 1 stored is visible)
            const double d0 = sx[lane], d1 = sx[64 + lane];
            const double dtop = wave_max_f64(fmax(d0, d1));
            if (!(dtop > dfloor)) break;                          // (uniform)
            const int f0 = wave_first(d0 >= dtop * PIV_BAND), f1 = wave_first(d1 >= dtop * PIV_BAND);
            const int pv = f0 >= 0 ? f0 : 64 + f1;
            const double dmax = sx[pv];
            const double gcol = (mine && !done) ? G[(size_t)pv * ldg + tid] : 0.0;     // in flight behind the factor's columns
            const double lpp = sqrt(dmax), inv = 1.0 / lpp;
            double dot0 = 0.0, dot1 = 0.0, dot2 = 0.0, dot3 = 0.0;
            // the products L(i, 0:s) . L(pv, 0:s): waves 0, 1 own rows 0 .. 127; waves 2, 3 shadow the same rows and take every other
            // group of 8 columns off them (their partial sums cross in `sco`, one more barrier) -- all four SIMDs work
            const int half = tid >> 7, rowi = tid & 127;
            if (64 * (wave & 1) < k) {                            // (wave-uniform: every lane of a wave with rows takes part --
                // v_readlane reads lanes whatever the exec mask says, so the lanes beyond k must hold the row as well)
                // the pivot's row of the factor, one entry per lane (two registers: s <= 128), handed out by v_readlane below:
                // the shared LDS pipe then carries ONE read per multiply-add (each thread's own row), not two
                const double rp0 = lane < s ? Lf[(size_t)lane * ldl + pv] : 0.0;
                const double rp1 = 64 + lane < s ? Lf[(size_t)(64 + lane) * ldl + pv] : 0.0;
                const int tidc = rowi < k ? rowi : 0;             // (lanes beyond k read row 0: in range, result unused)
                auto pivot_entry = [&](int t) -> double {         // (t wave-uniform)
                    const double r = t < 64 ? rp0 : rp1;
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r), t & 63),
                                            __builtin_amdgcn_readlane(__double2loint(r), t & 63));
                };
                int t0 = 8 * half;
                for (; t0 + 8 <= s; t0 += 16) {
                    double av[8];
#pragma unroll
                    for (int u_ = 0; u_ < 8; ++u_) av[u_] = Lf[(size_t)(t0 + u_) * ldl + tidc];
#pragma unroll
                    for (int u_ = 0; u_ < 8; u_ += 4) {
                        dot0 = fma(av[u_], pivot_entry(t0 + u_), dot0);
                        dot1 = fma(av[u_ + 1], pivot_entry(t0 + u_ + 1), dot1);
                        dot2 = fma(av[u_ + 2], pivot_entry(t0 + u_ + 2), dot2);
                        dot3 = fma(av[u_ + 3], pivot_entry(t0 + u_ + 3), dot3);
                    }
                }
                if (t0 < s) {                                     // this half's last, partial group: columns past s - 1 read column
                    double av[8];                                 // s - 1 (finite) against a pivot entry of 0
#pragma unroll
                    for (int u_ = 0; u_ < 8; ++u_) { const int t = t0 + u_; av[u_] = Lf[(size_t)(t < s ? t : s - 1) * ldl + tidc]; }
#pragma unroll
                    for (int u_ = 0; u_ < 8; u_ += 2) {
                        dot0 = fma(av[u_], pivot_entry(t0 + u_), dot0);
                        dot1 = fma(av[u_ + 1], pivot_entry(t0 + u_ + 1), dot1);
                    }
                }
            }
            dot0 = (dot0 + dot2) + (dot1 + dot3);
            if (half == 1 && rowi < k) sco[rowi] = dot0;
            __syncthreads();                                      // C
            if (mine) dot0 += sco[tid];
            dot1 = 0.0;
            const double lis = (tid == pv) ? lpp : (gcol - (dot0 + dot1)) * inv;
            if (mine && !done) Lf[(size_t)s * ldl + tid] = lis;
            if (tid == pv) { s_bc[0] = bvec * inv; sord[s] = pv; }     // forward substitution: w_s = b_pv / lpp
            __threadfence_block();
            __syncthreads();                                      // B
            const double ws = s_bc[0];
            if (mine && !done && tid != pv) { diag = fma(-lis, lis, diag); bvec = fma(-lis, ws, bvec); }
            if (tid == pv) { done = true; mystep = s; bvec = ws; myinv = inv; }
            rank++;
        }
        __syncthreads();
        double y = 0.0;
        for (int s = rank - 1; s >= 0; --s) {                     // (column-oriented; the broadcast slot alternates: one barrier a step)
            const int pvs = sord[s];
            if (tid == pvs) { y = bvec * myinv; s_bc[1 + (s & 1)] = y; }
            __syncthreads();
            const double ys = s_bc[1 + (s & 1)];
            if (mine && mystep >= 0 && mystep < s) bvec = fma(-Lf[(size_t)mystep * ldl + pvs], ys, bvec);
        }
        __syncthreads();
        y_out = y; pivoted = mystep >= 0;
    };
    auto lsq_on = [&](bool inP, double &y_out, bool &pivoted) {
        if (lds_factor) lsq_impl(std::true_type{}, inP, y_out, pivoted);
        else lsq_impl(std::false_type{}, inP, y_out, pivoted);
    };
    // r = A_bar y - q~: thread <-> entry t (two per thread beyond 256); the residual's square sum over the block
    double rt[2];
    auto residual = [&](double ysc) -> double {
        if (mine) sco[tid] = ysc * dcol * ssgv[tid];              // coefficient of the UNSCALED, unsigned row
        __syncthreads();
        double ss = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = tid + WTPB * h;
            rt[h] = 0.0;
            if (t < n) {
                double s0 = -qt[h], s1 = 0.0;
                int c = 0;
                for (; c + 8 <= k; c += 8) {
                    double v[8];
#pragma unroll
                    for (int u_ = 0; u_ < 8; ++u_) v[u_] = A_[(size_t)t * m + srow[c + u_]];
#pragma unroll
                    for (int u_ = 0; u_ < 8; u_ += 2) { s0 = fma(v[u_], sco[c + u_], s0); s1 = fma(v[u_ + 1], sco[c + u_ + 1], s1); }
                }
                for (; c < k; ++c) s0 = fma(A_[(size_t)t * m + srow[c]], sco[c], s0);
                rt[h] = s0 + s1;
                ss += rt[h] * rt[h];
            }
        }
        return block_sum_f64(ss, red);
    };

    double y = 0.0; bool piv = false;
    bool solved_fast = false;
    if (fast_chol) {                                               // (uniform)
        if (tid < 16 * T) sco[tid] = mine ? rhs : 0.0;
        if (tid == 0) s_pv = 0;
        __syncthreads();
        if (tc_factor(dynp, T, &s_pv, 1e-9, tid)) {
            if (wave == 0) tc_solve(dynp, sco, T, lane);
            __syncthreads();
            y = mine ? sco[tid] : 0.0; piv = mine;
            solved_fast = true;
        }
        __syncthreads();
    }
    if (!solved_fast) lsq_on(mine, y, piv);
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (tid < npn && !(y * dcol > -a.tol)) s_flag = 1;           // :119 signs (y scaled back)
    double res;
    if (solved_fast) {
        // y solves the normal equations G y = b of the full column set, so |A_bar y - q~|^2 = |q~|^2 - y'b: no third pass over Ad
        // for the accept test (rounding: 1e-16 |q~|^2, i.e. a residual norm resolved to ~1e-7 against the tolerance 1e-4); the
        // bounded least squares below forms its residuals explicitly
        const double qq = block_sum_f64(qt[0] * qt[0] + qt[1] * qt[1], red);
        const double yb = block_sum_f64(mine ? y * rhs : 0.0, red);
        res = qq > yb ? qq - yb : 0.0;
        if (mine) sco[tid] = y * dcol * ssgv[tid];                // (what residual() leaves there: the unscaled multipliers)
    } else res = residual(y);
    __syncthreads();
    if (!s_flag && sqrt(res) <= a.tol) {
        for (int r = tid; r < m; r += WTPB) lam[r] = 0.0;
        __syncthreads();
        if (mine) lam[srow[tid]] = sco[tid];                      // :120-123
        if (tid == 0) { a.solution[b] = 1; a.path[b] = 2; }
        return;
    }
    // ---- :129-137  bounded least squares (verify_node32's active-set iteration; G's principal submatrices are re-factored)
    {
        const bool cons = tid < npn;
        bool inP = mine && piv, blocked = false, failed = false;
        double ycur = 0.0, sl = y;
        int iters = 0;
        const int cap = 3 * k + 12;
        for (;;) {
            for (;;) {
                const bool bad = inP && cons && !(sl > 0.0);
                if (tid == 0) s_flag = 0;
                __syncthreads();
                if (bad) s_flag = 1;
                __syncthreads();
                if (!s_flag) { ycur = inP ? sl : 0.0; break; }
                const double ratio = bad ? ycur / (ycur - sl) : QINF;
                const double mymin = (bad && ratio == ratio) ? ratio : (bad ? 0.0 : QINF);
                const double alpha = -block_max_f64(-mymin, red);
                if (inP) ycur = fma(alpha, sl - ycur, ycur);
                if (bad && (!(ratio == ratio) || ratio <= alpha)) { inP = false; ycur = 0.0; }
                if (++iters > cap) { failed = true; break; }
                bool pv2;
                __syncthreads();
                lsq_on(inP, sl, pv2);
                if (inP && !pv2) { inP = false; ycur = 0.0; }
            }
            if (failed) break;
            res = residual(ycur);
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int t = tid + WTPB * h; if (t < n) sx[t] = rt[h]; }
            __syncthreads();
            double wgr = 0.0;
            if (mine) {
                double w0 = 0.0, w1 = 0.0;
                int t = 0;
                for (; t + 2 <= n; t += 2) {
                    w0 = fma(A_[(size_t)t * m + myrow], sx[t], w0);
                    w1 = fma(A_[(size_t)(t + 1) * m + myrow], sx[t + 1], w1);
                }
                if (t < n) w0 = fma(A_[(size_t)t * m + myrow], sx[t], w0);
                wgr = -(w0 + w1) * dcol * ssgv[tid];              // scaled, signed column . (q~ - A_bar y)
            }
            const double gscale = block_max_f64(mine ? fabs(rhs) : 0.0, red);
            const bool cand = mine && cons && !inP && !blocked && wgr > 1e-11 * (gscale > 1.0 ? gscale : 1.0);
            const double wmax = block_max_f64(cand ? wgr : -1.0, red);
            if (!(wmax > 0.0)) break;                             // optimal
            if (tid == 0) s_pv = 0x7fffffff;
            __syncthreads();
            if (cand && wgr == wmax) atomicMin(&s_pv, tid);
            __syncthreads();
            const int enter = s_pv;
            if (tid == enter) inP = true;
            if (++iters > cap) { failed = true; break; }
            bool pv2;
            lsq_on(inP, sl, pv2);
            if (inP && !pv2) { inP = false; if (tid == enter) blocked = true; }
            if (tid == enter) s_bc[0] = sl;
            __syncthreads();
            const double s_enter = s_bc[0];
            __syncthreads();
            if (!(s_enter > 0.0)) {
                if (tid == enter) { inP = false; blocked = true; }
                lsq_on(inP, sl, pv2);
                if (inP && !pv2) inP = false;
            }
        }
        for (int r = tid; r < m; r += WTPB) lam[r] = 0.0;
        if (failed) {                                             // :143-145
            if (tid == 0) { a.solution[b] = 0; a.path[b] = 5; }
            return;
        }
        __syncthreads();
        const bool ok = sqrt(res) <= 1e-4;                        // :138 (res: the residual at ycur, from the last outer test)
        if (mine) lam[srow[tid]] = sco[tid];
        if (tid == 0) { a.solution[b] = ok ? 1 : 0; a.path[b] = ok ? 3 : 4; }
    }
}

__global__ __launch_bounds__(WTPB) void verify_wide_stage1(VerifyArgs a)
{
    const int n = a.n, m = a.m, p = a.p;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (a.gate && a.path[b] != -2) return;
    __shared__ double sx[WMAX], sqt[WMAX], ssg[WMAX], sdiag[WMAX], sb[WMAX], sy[WMAX], slv[WMAX];
    __shared__ int srow[WMAX], sstep[WMAX], scls[WMAX], sord[WMAX];
    __shared__ double red[4];
    __shared__ int s_flag, s_k, s_np, s_nn, s_pv;

    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *B_ = a.B + (size_t)b * m * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double *lam = a.lambda + (size_t)b * m;
    double *G = a.sG + (size_t)b * m * m;                  // Gram block / factor, leading dimension m

    for (int i = tid; i < n; i += WTPB) sx[i] = a.xd[(size_t)b * n + i];
    if (tid == 0) s_flag = 0;
    __syncthreads();
    // :58-60  q~ (thread i <-> row i: coalesced columns of Qd)
    for (int i = tid; i < n; i += WTPB) {
        double qt = a.qd[(size_t)b * n + i];
        for (int j = 0; j < n; ++j) qt = fma(Q_[(size_t)j * n + i], sx[j], qt);
        for (int k = 0; k < p; ++k) qt = fma(R_[(size_t)k * n + i], w_[k], qt);
        sqt[i] = qt;
    }
    // :84 ax, :86 feasibility, :98-103 classes
    for (int r = tid; r < m; r += WTPB) {
        double ax = 0.0;
        for (int j = 0; j < n; ++j) ax = fma(A_[(size_t)j * m + r], sx[j], ax);
        for (int k = 0; k < p; ++k) ax = fma(B_[(size_t)k * m + r], w_[k], ax);
        const double lr = a.l[(size_t)b * m + r], ur = a.u[(size_t)b * m + r];
        lam[r] = 0.0;
        if (!(lr - 1e-3 <= ax && ax - 1e-3 <= ur)) s_flag = 1;
        scls[r] = ((ax < lr + 1e-2) ? 1 : 0) | ((ax > ur - 1e-2) ? 2 : 0);
    }
    __syncthreads();
    if (s_flag) { if (tid == 0) { a.solution[b] = 0; a.path[b] = 0; } return; }
    if (m == 0) {   // :91-96
        double s = 0.0;
        for (int i = tid; i < n; i += WTPB) s += sqt[i] * sqt[i];
        s = block_sum_f64(s, red);
        if (tid == 0) { a.solution[b] = sqrt(s) <= a.tol ? 1 : 0; a.path[b] = 1; }
        return;
    }
    // active columns in the order [pos | neg | both] (:114), rows ascending within a class
    if (tid == 0) {
        int c = 0, np = 0, nn = 0;
        for (int pass = 1; pass <= 3; ++pass)
            for (int r = 0; r < m; ++r)
                if (scls[r] == pass) { srow[c] = r; ssg[c] = pass == 2 ? -1.0 : 1.0; ++c; if (pass == 1) ++np; else if (pass == 2) ++nn; }
        s_k = c; s_np = np; s_nn = nn;
    }
    __syncthreads();
    const int k = s_k, npn = s_np + s_nn;
    // Gram entries (ascending fma chains over t, as in the small kernel) and the right-hand side
    for (int e = tid; e < k * k; e += WTPB) {
        const int c1 = e % k, c2 = e / k;
        const int r1 = srow[c1], r2 = srow[c2];
        double s = 0.0;
        for (int t = 0; t < n; ++t) s = fma(A_[(size_t)t * m + r1], A_[(size_t)t * m + r2], s);
        G[(size_t)c2 * m + c1] = s * ssg[c1] * ssg[c2];
    }
    for (int c = tid; c < k; c += WTPB) {
        double rhs = 0.0;
        const int r = srow[c];
        for (int t = 0; t < n; ++t) rhs = fma(A_[(size_t)t * m + r], sqt[t], rhs);
        sb[c] = rhs * ssg[c];
        sstep[c] = -1;
    }
    __syncthreads();
    for (int c = tid; c < k; c += WTPB) sdiag[c] = G[(size_t)c * m + c];
    double dsc = 0.0;
    for (int c = tid; c < k; c += WTPB) dsc = fmax(dsc, G[(size_t)c * m + c]);
    const double dscale = block_max_f64(dsc, red);
    int rank = 0;
    for (int s = 0; s < k; ++s) {
        double dm = -1.0;
        for (int c = tid; c < k; c += WTPB) if (sstep[c] < 0) dm = fmax(dm, sdiag[c]);
        const double dmax = block_max_f64(dm, red);
        if (!(dmax > 1e-12 * (dscale > 1.0 ? dscale : 1.0))) break;
        if (tid == 0) s_pv = WMAX;
        __syncthreads();
        for (int c = tid; c < k; c += WTPB) if (sstep[c] < 0 && sdiag[c] >= dmax * PIV_BAND) atomicMin(&s_pv, c);     // lowest such column
        __syncthreads();
        const int pv = s_pv;
        const double lpp = sqrt(sdiag[pv]);
        for (int c = tid; c < k; c += WTPB) {
            double lis = 0.0;
            if (sstep[c] < 0) lis = c == pv ? lpp : G[(size_t)pv * m + c] / lpp;
            slv[c] = (sstep[c] >= 0 || c == pv) ? 0.0 : lis;
            sy[c] = lis;                                  // (sy doubles as this step's L(:, s) until the back substitution)
        }
        __syncthreads();
        const double ws = sb[pv] / lpp;
        for (int c = tid; c < k; c += WTPB) {
            if (sstep[c] < 0 && c != pv) {
                const double lis = sy[c];
                for (int j = 0; j < k; ++j) {
                    const double lj = slv[j];
                    if (lj != 0.0) G[(size_t)j * m + c] = fma(-lis, lj, G[(size_t)j * m + c]);
                }
                sdiag[c] = G[(size_t)c * m + c];
                sb[c] = fma(-lis, ws, sb[c]);
            }
        }
        __syncthreads();
        for (int c = tid; c < k; c += WTPB) if (sstep[c] < 0) G[(size_t)pv * m + c] = sy[c];      // L(:, s) into the freed column
        __syncthreads();
        if (tid == 0) { sstep[pv] = s; sb[pv] = ws; sord[s] = pv; }
        rank++;
        __syncthreads();
    }
    // back substitution  L' y = w  over the pivoted columns, last step first
    for (int c = tid; c < k; c += WTPB) sy[c] = 0.0;
    __syncthreads();
    for (int s = rank - 1; s >= 0; --s) {
        const int pv = sord[s];
        double term = 0.0;
        for (int c = tid; c < k; c += WTPB) if (sstep[c] > s) term += G[(size_t)pv * m + c] * sy[c];
        const double acc = block_sum_f64(term, red);
        if (tid == 0) sy[pv] = (sb[pv] - acc) / G[(size_t)pv * m + pv];
        __syncthreads();
    }
    // :119  sign and residual tests
    if (tid == 0) s_flag = 0;
    __syncthreads();
    for (int c = tid; c < npn; c += WTPB) if (!(sy[c] > -a.tol)) s_flag = 1;
    for (int c = tid; c < k; c += WTPB) slv[c] = sy[c] * ssg[c];
    __syncthreads();
    double res = 0.0;
    for (int i = tid; i < n; i += WTPB) {
        double s = -sqt[i];
        for (int c = 0; c < k; ++c) s = fma(A_[(size_t)i * m + srow[c]], slv[c], s);
        res += s * s;
    }
    res = block_sum_f64(res, red);
    const bool ok = !s_flag && sqrt(res) <= a.tol;
    if (ok) {
        for (int c = tid; c < k; c += WTPB) lam[srow[c]] = (ssg[c] < 0.0) ? -sy[c] : sy[c];       // :120-123
        if (tid == 0) { a.solution[b] = 1; a.path[b] = 2; }
        return;
    }
    // ---- :129-137  bounded least squares as a box-AVI in lambda, handed to the large-item AVI kernel ---
    __syncthreads();
    for (int e = tid; e < m * m; e += WTPB) {
        const int i = e % m, j = e / m;
        double s = 0.0;
        for (int t = 0; t < n; ++t) s = fma(A_[(size_t)t * m + i], A_[(size_t)t * m + j], s);
        G[(size_t)j * m + i] = s;
    }
    for (int r = tid; r < m; r += WTPB) {
        double s = 0.0;
        for (int t = 0; t < n; ++t) s = fma(A_[(size_t)t * m + r], sqt[t], s);
        a.sq[(size_t)b * m + r] = -s;
        a.slb[(size_t)b * m + r] = (scls[r] & 2) ? -QINF : 0.0;   // :129-131
        a.sub[(size_t)b * m + r] = (scls[r] & 1) ? QINF : 0.0;    // :132-134
        a.sz[(size_t)b * m + r] = 0.0;
    }
    if (tid == 0) { a.solution[b] = 0; a.path[b] = -1; }
}

__global__ __launch_bounds__(WTPB) void verify_wide_stage2(VerifyArgs a, const int32_t *avi_status)
{
    const int n = a.n, m = a.m, p = a.p;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (a.path[b] != -1) return;
    double *lam = a.lambda + (size_t)b * m;
    __shared__ double sl_[WMAX], sx[WMAX], red[4];
    if (avi_status[b] != QPN_SUCCESS) {   // :143-145
        for (int r = tid; r < m; r += WTPB) lam[r] = 0.0;
        if (tid == 0) { a.solution[b] = 0; a.path[b] = 5; }
        return;
    }
    for (int r = tid; r < m; r += WTPB) { sl_[r] = a.sz[(size_t)b * m + r]; lam[r] = sl_[r]; }
    for (int i = tid; i < n; i += WTPB) sx[i] = a.xd[(size_t)b * n + i];
    __syncthreads();
    const double *Q_ = a.Qd + (size_t)b * n * n;
    const double *A_ = a.Ad + (size_t)b * m * n;
    const double *R_ = a.R + (size_t)b * n * p;
    const double *w_ = a.w + (size_t)b * (size_t)a.stride_w;
    double res = 0.0;
    for (int i = tid; i < n; i += WTPB) {
        double qt = a.qd[(size_t)b * n + i];
        for (int j = 0; j < n; ++j) qt = fma(Q_[(size_t)j * n + i], sx[j], qt);
        for (int k = 0; k < p; ++k) qt = fma(R_[(size_t)k * n + i], w_[k], qt);
        double s = -qt;
        for (int r = 0; r < m; ++r) s = fma(A_[(size_t)i * m + r], sl_[r], s);
        res += s * s;
    }
    res = block_sum_f64(res, red);
    if (tid == 0) {
        const bool ok = sqrt(res) <= 1e-4;   // :138
        a.solution[b] = ok ? 1 : 0;
        a.path[b] = ok ? 3 : 4;
    }
}

} // namespace

int qpn_verify_max_dim() { return WMAX; }

hipError_t qpn_launch_verify_nodes(int32_t batch, int32_t n, int32_t m, int32_t p,
                                   const double *Qd, const double *R, const double *qd,
                                   const double *Ad, const double *B, const double *l,
                                   const double *u, const double *xd, const double *w,
                                   int64_t stride_w, double tol, int32_t *solution, double *lambda,
                                   int32_t *path, double *sG, double *sq, double *slb, double *sub,
                                   double *sz, double *sres, int32_t *sst, hipStream_t stream, double *wbig, double *gws)
{
    if (batch <= 0) return hipSuccess;
    VerifyArgs a{};
    a.batch = batch; a.n = n; a.m = m; a.p = p;
    a.Qd = Qd; a.R = R; a.qd = qd; a.Ad = Ad; a.B = B; a.l = l; a.u = u; a.xd = xd; a.w = w;
    a.stride_w = stride_w; a.tol = tol; a.solution = solution; a.lambda = lambda; a.path = path;
    a.sG = sG; a.sq = sq; a.slb = slb; a.sub = sub; a.sz = sz;
    if (n > 64 || m > 64) {
        // wide nodes: verify_wide_node (up to 128 active rows, every path inside the workgroup), then round 1's kernels over what it
        // flagged -2 (their bounded-LSQ fallback is a large box-AVI, N = m, on the large-item kernel)
        if (m >= 1 && gws) {
            static QpnPerDeviceOnce attr_once;
            const int dv = attr_once.device();
            if (!attr_once.done[dv]) {
                hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(verify_wide_node), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
                if (e0 != hipSuccess) return e0;
                attr_once.done[dv] = true;
            }
            const int wp = m <= 256 ? 16 : 8;
            // the panels, and room for a k x k factor (k <= min(m, 128)) up to what two workgroups per CU allow
            size_t dyn = (size_t)2 * wp * (size_t)(m | 1) * sizeof(double);
            const size_t km = (size_t)(m < 128 ? m : 128), want = km * km * sizeof(double), cap = 71 * 1024;
            if (dyn < (want < cap ? want : cap)) dyn = want < cap ? want : cap;
            hipLaunchKernelGGL(verify_wide_node, dim3((unsigned)batch), dim3(WTPB), dyn, stream, a, gws, wp, (int)(dyn / sizeof(double)), (int *)nullptr, 0);
            hipError_t e1 = hipGetLastError();
            if (e1 != hipSuccess) return e1;
            a.gate = 1;
        }
        hipLaunchKernelGGL(verify_wide_stage1, dim3((unsigned)batch), dim3(WTPB), 0, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || m == 0) return e;
        AviBatchArgs s{};
        s.batch = batch; s.N = m; s.M = sG; s.strideM = (int64_t)m * m; s.q = sq; s.l = slb; s.u = sub;
        s.kind = nullptr; s.stride_kind = 0; s.z = sz; s.status = sst; s.resid = sres; s.pivots = nullptr;
        s.active = nullptr; s.check_tol = 1e-6; s.piv_tol = 1e-11; s.feas_tol = 1e-12; s.comp_tol = 1e-2;
        s.max_pivots = 0; s.only_if = path; s.only_if_value = -1;
        e = m > 64 ? qpn_launch_avi_solve_big(s, wbig, stream) : (s.scan = 1, qpn_launch_avi_solve(s, stream));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(verify_wide_stage2, dim3((unsigned)batch), dim3(WTPB), 0, stream, a, (const int32_t *)sst);
        return hipGetLastError();
    }
    if (n <= 32 && m <= 32) {
        // ONE launch: the bounded least-squares fallback runs inside the node's wavefront (verify_node32)
        if (n == 32 && m == 32) hipLaunchKernelGGL(verify_node32<true>, dim3((unsigned)batch), dim3(WAVE), 0, stream, a);
        else hipLaunchKernelGGL(verify_node32<false>, dim3((unsigned)batch), dim3(WAVE), 0, stream, a);
        return hipGetLastError();
    }
    // 33 .. 64: verify_node64 (up to 32 active rows: the usual case), then the older kernels over what it flagged -2
    if (m >= 1) {
        hipLaunchKernelGGL(verify_node64, dim3((unsigned)batch), dim3(WAVE), 0, stream, a);
        a.gate = 1;
        if (gws) {
            // the nodes it flagged (more than 32 active rows: a handful in thousands) go to the workgroup kernel first -- ~35 us for
            // such a node instead of ~130 us on the one-wavefront kernels of round 1, which bound the whole call's tail
            static QpnPerDeviceOnce attr_once;
            const int dv = attr_once.device();
            if (!attr_once.done[dv]) {
                hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(verify_wide_node), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
                if (e0 != hipSuccess) return e0;
                attr_once.done[dv] = true;
            }
            const size_t mp16 = (size_t)((m + 15) & ~15);
            int *ctr = reinterpret_cast<int *>(gws + (size_t)QPN_VERIFY_MID_SLOTS * 2 * mp16 * mp16);
            hipError_t e0 = hipMemsetAsync(ctr, 0, 4, stream);
            if (e0 != hipSuccess) return e0;
            size_t dyn = (size_t)2 * 16 * (size_t)(m | 1) * sizeof(double);
            const size_t want = (size_t)m * m * sizeof(double);
            if (dyn < want) dyn = want;
            hipLaunchKernelGGL(verify_wide_node, dim3((unsigned)batch), dim3(WTPB), dyn, stream, a, gws, 16, (int)(dyn / sizeof(double)), ctr, QPN_VERIFY_MID_SLOTS);
            e0 = hipGetLastError();
            if (e0 != hipSuccess) return e0;
        }
    }
    hipLaunchKernelGGL(verify_stage1<65>, dim3((unsigned)batch), dim3(WAVE), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || m == 0) return e;
    AviBatchArgs s{};
    s.batch = batch; s.N = m; s.M = sG; s.strideM = (int64_t)m * m; s.q = sq; s.l = slb; s.u = sub;
    s.kind = nullptr; s.stride_kind = 0; s.z = sz; s.status = sst; s.resid = sres; s.pivots = nullptr;
    s.active = nullptr; s.check_tol = 1e-6; s.piv_tol = 1e-11; s.feas_tol = 1e-12; s.comp_tol = 1e-2;
    s.max_pivots = 0; s.only_if = path; s.only_if_value = -1; s.scan = 1;     // compact fallback launch
    e = qpn_launch_avi_solve(s, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(verify_stage2, dim3((unsigned)batch), dim3(WAVE), 0, stream, a, (const int32_t *)sst);
    return hipGetLastError();
}
