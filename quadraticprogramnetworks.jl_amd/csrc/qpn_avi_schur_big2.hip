// qpn_avi_schur_big2.hip -- the blocked matrix-core crash for LARGE node records (64 < n <= 256, m <= 256: BASELINE config 5),
// straight from the records, gfx950.
//
// qpn_avi_schur_big.hip works on an ASSEMBLED M (2 MB written and read back per node at n = m = 256) and eliminates the top
// half [H | C | g] with rank-16 / rank-32 passes over an HBM workspace: 16 + 1 passes over ~1 MB per node by four wavefronts
// with two tiles in flight each -- 2.8 ms of a 5.3 ms sweep, neither the matrix cores (23 % busy) nor HBM (2.4 TB/s) full.
// Here:
//   * nothing is assembled: pass 0 reads Qd and Ad from the records (src/avi.jl:205-251: H = Qd, C = -Ad', g = qd + R w),
//     max |M| falls out of that pass (the pivots accepted before it is known are re-checked once it is);
//   * one workgroup of 16 wavefronts per node, wavefront I owns ROW TILE I of the top half for the whole elimination;
//   * RANK-64 block pivots: the 64 x 64 pivot block is inverted in LDS (block Gauss-Jordan on 16 x 16 tiles, the diagonal
//     tiles by an in-register LU in wave 0 with the same per-pivot threshold as everywhere else), every wave forms its 16 x 64
//     piece of U' = (U - E) P^-1 as the TRANSPOSED product (P^-T U'), whose accumulator layout IS the A-operand layout of the
//     update -- U' lives in 32 registers, never in LDS;
//   * the raw pivot rows (B operands, shared by all waves) cross LDS in chunks of four column tiles, double-buffered: one
//     workgroup barrier per chunk; a tile of the top half is read and written ONCE per pass by its owner: 4 passes at
//     n = 256 instead of 8 (paired) / 16, dead column tiles are never touched;
//   * S = -Ad W and c = b - Ad h (src/avi.jl:305-377 in Schur form) with the same ownership: wave I holds row tile I of S,
//     W crosses LDS in 64 x 64 blocks.
// Outputs are those of schur_big_stage_a (workspace views of qpn_internal.h: W | h in the row-major top half, S, c, bounds,
// cold start), so the delayed-update Lemke kernel runs unchanged behind it; schur_big2_finish is the read-back and post-check
// (src/avi.jl:71-76, :148-156; src/avi_solutions.jl:511-562) on the records instead of an assembled M.
// Declined nodes (a pivot below the threshold, an equality row) keep status -1 and take the general path in gated launches.
#include "qpn_internal.h"
#include <type_traits>
#include <cstdlib>

#define QINF __builtin_huge_val()

namespace {

constexpr int TPB2 = 1024;              // 16 wavefronts
constexpr int LDP = 80;                 // row stride of the pivot block / its inverse (== 16 mod 32: rows lq, lq + 1 on other banks)
constexpr int LDD = 17;                 // row stride of the inverted diagonal tile
constexpr int JC = 4;                   // column tiles per staged block of W in the S product
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)
#define MFMA_NEGA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 1)      // D = C - A B

// LDS map of the elimination (doubles)
constexpr int MC = 13;                              // column tiles per mega-chunk of pivot rows
constexpr int OFF_P = 0;                            // pivot block, inverted in place                      [64][LDP]
constexpr int OFF_D = OFF_P + 64 * LDP;             // inverse of the current diagonal tile                [16][LDD] (+ pad)
constexpr int OFF_V = OFF_D + 288;                  // raw pivot rows of a mega-chunk, as tiles            [MC][4][256]
constexpr int OFF_G = OFF_V + MC * 4 * 256;         // g = qd + R w (pass 0 from the records)              [256]
constexpr int OFF_RED = OFF_G + 256;                // block reductions, flags                             [32]
constexpr int LDS_DOUBLES = OFF_RED + 32;           // 19 008 doubles = 148.5 KB: one workgroup per CU
constexpr int LDS_SPROD = 2 * 16 * 256 + 256;       // the S product: two blocks of W (4 block rows x JC tiles) + b

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

__device__ __forceinline__ int pad16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ double max_abs(double a, double b)      // max(a, |b|)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double rcp64(double x)      // two Newton steps on v_rcp_f64: <= 1 ulp of 1 / x for normal x
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Inverse of a 16 x 16 tile (LDS, row stride ldp) by in-place Gauss-Jordan without pivoting, in the registers of ONE wavefront:
// lane i < 16 holds row i; step s broadcasts row s with v_readlane and every row takes  a_i <- a_i - (f_i / p) a_s  with
// f_i = a_is - [i = s] (the pivot row carries p - 1, so that the same update turns it into a_s / p: no select), column s
// becomes -(f_i / p) + [i = s].  The pivots are those of an LU without pivoting (the Schur complements' diagonals): each must
// pass `thr`; the smallest |pivot| is handed back.  Result in D (row stride LDD).  false = a pivot failed.
__device__ bool invert16(const double *P, int ldp, double *D, double thr, double &minpiv, int lane)
{
    double ar[16];
    const int li = lane & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) ar[k] = P[li * ldp + k];
    bool ok = true;
    double mp = QINF;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
        const double piv = readlane_f64(ar[s_], s_);
        ok = ok && (fabs(piv) >= thr);
        mp = fmin(mp, fabs(piv));
        const double r = rcp64(piv);
        const double e = (li == s_) ? 1.0 : 0.0;
        const double g = (ar[s_] - e) * r;
#pragma unroll
        for (int j2 = 0; j2 < 16; ++j2)
            if (j2 != s_) ar[j2] = fma(-g, readlane_f64(ar[j2], s_), ar[j2]);
        ar[s_] = e - g;
    }
    if (lane < 16) {
#pragma unroll
        for (int j2 = 0; j2 < 16; ++j2) D[lane * LDD + j2] = ar[j2];
    }
    minpiv = fmin(minpiv, mp);
    return ok;
}

// The top half lives in the workspace TILE BY TILE: tile (I, J) is 256 contiguous doubles at ((I nct + J) * 4 + g) * 64 + lane
// -- the accumulator layout of v_mfma_f64_16x16x4_f64 (element (row 4 g + lq, column lc) in register g of lane 16 lq + lc), so a
// tile moves with four 512-byte loads, the raw pivot rows ARE B operands as they stand (k-block s of block row s / 4 is
// register s % 4 of the same lane: the chunks cross LDS as plain copies, read back conflict-free), and a node's whole top half
// is one contiguous megabyte (a row-major top half made every tile 16 rows 4 KB apart).

// ---- K0: the padded top half, tile by tile, from the records (src/avi.jl:205-251: H = Qd, C = -Ad', g = qd + R w); max |M| -----
__global__ __launch_bounds__(256) void schur_big2_convert(AviBatchArgs a, SchurBigWs w)
{
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, lc = lane & 15, lq = lane >> 4, wave = tid >> 6;
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    const int nrt = pad16(n) >> 4, mct = pad16(m) >> 4, nct = nrt + mct + 1;
    __shared__ double sG[256];
    __shared__ double red[4];
    const double *Q_ = a.nd.Qd + (size_t)b * n * n;
    const double *A_ = a.nd.Ad + (size_t)b * m * n;
    const double *R_ = a.nd.R + (size_t)b * n * np_;
    const double *w_ = a.nd.w + (size_t)b * (size_t)a.nd.stride_w;
    double *const Tt = w.Tt + (size_t)b * (size_t)w.tt_stride;
    for (int i = tid; i < n; i += 256) {
        double s = a.nd.qd[(size_t)b * n + i];
        for (int k = 0; k < np_; ++k) s = fma(R_[(size_t)k * n + i], w_[k], s);
        sG[i] = s;
    }
    __syncthreads();
    double mabs = 0.0;
    // element (g, lane) of the initial tile t = I nct + J (clamped addresses, the value selected afterwards: no branches)
    auto elem = [&](int t, int g) -> double {
        const int I = t / nct, J = t - I * nct;
        const int r = 16 * I + 4 * g + lq;
        if (J < nrt) {
            const int c = 16 * J + lc;
            const bool ok = r < n && c < n;
            const double q = Q_[ok ? c * n + r : 0];
            return ok ? q : (r == c ? 1.0 : 0.0);
        }
        if (J < nrt + mct) {
            const int k = 16 * (J - nrt) + lc;
            const bool ok = r < n && k < m;
            const double q = A_[ok ? r * m + k : 0];
            return ok ? -q : 0.0;
        }
        const double q = sG[r < n ? r : 0];
        return (lc == 0 && r < n) ? q : 0.0;
    };
    const int nt = nrt * nct;
    for (int t0 = 4 * wave; t0 < nt; t0 += 16) {        // four tiles per wave and round: sixteen loads in flight
        double v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[q][g] = t0 + q < nt ? elem(t0 + q, g) : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mabs = max_abs(mabs, v[q][g]);
                if (t0 + q < nt) Tt[((size_t)(t0 + q) << 8) + g * 64 + lane] = v[q][g];
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mabs, off, 64); mabs = o > mabs ? o : mabs; }
    if (lane == 0) red[wave] = mabs;
    __syncthreads();
    if (tid == 0) w.c[(size_t)b * N] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));       // (until the S product writes c)
}

// ---- K1: the elimination, rank-64 block pivots --------------------------------------------------------------------------
// MODE 0: every pass on the tiles K0 has laid out.  n, m multiples of 16 need no conversion pass -- MODE 1: pass 0 ALONE, reading
// Qd, Ad, qd, R, w straight from the records (the top half is first written by this pass's own update; max |M| falls out of it,
// and the pivots it accepted under the threshold's lower bound 1e-4 are looked at again), status -4 behind it; MODE 2: the
// remaining passes of those nodes.  (Two launches, not two code paths in one kernel: with both pass bodies inlined in one function
// the register allocator spills a hundred registers.)
template <int MODE>
__global__ __launch_bounds__(TPB2, 1) void schur_big2_eliminate(AviBatchArgs a, SchurBigWs w)
{
    constexpr bool RECORDS = MODE == 1;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    const int n = a.nd.n, m = a.nd.m;
    const int nrt = pad16(n) >> 4, mct = pad16(m) >> 4, nct = nrt + mct + 1;      // row tiles; column tiles: H | C | the g tile
    extern __shared__ __attribute__((aligned(32))) double sm[];
    double *const sP = sm + OFF_P, *const sD = sm + OFF_D, *const sV = sm + OFF_V;
    double *const sRed = sm + OFF_RED;
    int *const sFlag = reinterpret_cast<int *>(sRed + 24);
    double *const Tt = w.Tt + (size_t)b * (size_t)w.tt_stride;
    auto tile = [&](int I_, int J_) -> double * { return Tt + ((size_t)(I_ * nct + J_) << 8); };
#ifdef QPN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (MODE == 2) { if (a.status[b] != -4) return; }
    // equality rows are not taken here (their multiplier would have to be crashed in)
    int bad_row = 0;
    if constexpr (MODE != 2) { if (tid < m && a.nd.l[(size_t)b * m + tid] == a.nd.u[(size_t)b * m + tid]) bad_row = 1; }
    if (tid == 0) sFlag[0] = 1;
    if (__syncthreads_count(bad_row) > 0) { if (tid == 0) a.status[b] = -1; return; }

    double minpiv = QINF, mabs = 0.0;
    // the pivot threshold 1e-4 max(1, max |M|), max |M| from K0 (RECORDS: after pass 0)
    double thr = 1e-4;
    if constexpr (!RECORDS) { const double ms = w.c[(size_t)b * (n + m)]; thr = 1e-4 * (ms > 1.0 ? ms : 1.0); }
    const bool owner = wave < nrt;                      // this wavefront owns row tile `wave`
    const int I = wave;
    double *const sG = sm + OFF_G;                      // (RECORDS) g = qd + R w
    if constexpr (RECORDS) {
        if (tid < n) {
            const int np_ = a.nd.p;
            const double *R_ = a.nd.R + (size_t)b * n * np_;
            const double *w_ = a.nd.w + (size_t)b * (size_t)a.nd.stride_w;
            double s_ = a.nd.qd[(size_t)b * n + tid];
            for (int k = 0; k < np_; ++k) s_ = fma(R_[(size_t)k * n + tid], w_[k], s_);
            sG[tid] = s_;
        }
        __syncthreads();
    }

    auto pass = [&](auto first_c, int kb) -> bool {
        constexpr bool FIRST = decltype(first_c)::value;
        const double *Q_ = a.nd.Qd + (size_t)b * n * n;          // (FIRST only)
        const double *A_ = a.nd.Ad + (size_t)b * m * n;
        // element (g, lane) of the initial tile (I_, J_): H = Qd, C = -Ad', the g tile
        auto t0 = [&](int I_, int J_, int g) -> double {
            const int r = 16 * I_ + 4 * g + lq;
            if (J_ < nrt) return Q_[(16 * J_ + lc) * n + r];
            if (J_ < nrt + mct) return -A_[r * m + 16 * (J_ - nrt) + lc];
            const double q = sG[r];
            return lc == 0 ? q : 0.0;
        };
        const int bw = (nrt - 4 * kb) < 4 ? (nrt - 4 * kb) : 4;
        const int Jlo = 4 * kb + bw;                    // live column tiles: right of the block
        const int nlive = nct - Jlo;
        STAMP(0);   // setup / end of the previous pass
        // the raw pivot rows of `cnt` live column tiles from J0 on -> sV[jj][block row][256]: plain copies of tiles (the B
        // operands as they stand), one 2 KB tile per wave and round
        auto stage = [&](int J0, int cnt) {
            d4 v[4];                                    // (<= 4 tiles per wave: MC * 4 <= 64; all loads in flight, then the writes)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = wave + 16 * q, jj = t / bw, br = t - jj * bw;
                if (t < cnt * bw) {
                    if constexpr (FIRST) {
                        // entries 4 lane .. 4 lane + 3 of the tile: register g = lane >> 4, lanes 4 (lane & 15) ..: row 4 g + lq', columns cc ..
                        const int J = J0 + jj, g = lane >> 4, l4 = 4 * (lane & 15), r = 16 * (4 * kb + br) + 4 * g + (l4 >> 4), cc = l4 & 15;
                        if (J < nrt) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[q][e] = Q_[(16 * J + cc + e) * n + r];
                        } else if (J < nrt + mct) {
                            const d4 t_ = *reinterpret_cast<const d4 *>(A_ + r * m + 16 * (J - nrt) + cc);
                            v[q] = -t_;
                        } else { const double g_ = sG[r]; v[q] = d4{cc == 0 ? g_ : 0.0, 0.0, 0.0, 0.0}; }
                    } else v[q] = *reinterpret_cast<const d4 *>(tile(4 * kb + br, J0 + jj) + 4 * lane);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = wave + 16 * q, jj = t / bw, br = t - jj * bw;
                if (t < cnt * bw) *reinterpret_cast<d4 *>(sV + ((jj * 4 + br) << 8) + 4 * lane) = v[q];
            }
        };
        // ---- the pivot block, raw -> LDS (row-major); the first mega-chunk of pivot rows rides along
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + TPB2 * e;
            if constexpr (FIRST) {
                const int r = idx & 63, c = idx >> 6;       // (records: column-major, rows fastest)
                if (r < 16 * bw && c < 16 * bw) sP[r * LDP + c] = Q_[c * n + r];
            } else {
                const int tt = idx >> 8, ti = tt >> 2, tj = tt & 3, g = (idx >> 6) & 3, ln = idx & 63;
                if (ti < bw && tj < bw) sP[(16 * ti + 4 * g + (ln >> 4)) * LDP + 16 * tj + (ln & 15)] = tile(4 * kb + ti, 4 * kb + tj)[g * 64 + ln];
            }
        }
        stage(Jlo, nlive < MC ? nlive : MC);
        __syncthreads();
        STAMP(1);   // pivot block + first pivot rows in LDS
        // ---- P <- P^-1 in place: block Gauss-Jordan over the 16 x 16 tiles.  Step j:  D = P_jj^-1;  P_jk <- D P_jk (k != j);
        // P_ik <- P_ik - P_ij P_jk (i, k != j);  P_ij <- -P_ij D (i != j);  P_jj <- D.
        for (int j = 0; j < bw; ++j) {
            if (wave == 0) {
                const bool ok = invert16(sP + 16 * j * LDP + 16 * j, LDP, sD, thr, minpiv, lane);
                if (lane == 0 && !ok) sFlag[0] = 0;
            }
            __syncthreads();
            STAMP(5);   // (diagnostic) the diagonal tile's inverse (wave 0) + barrier
            if (sFlag[0] == 0) return false;
            if (wave < bw && wave != j) {               // row block j: tile (j, k = wave)
                const int k = wave;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc = MFMA(sD[lc * LDD + 4 * t + lq], sP[(16 * j + 4 * t + lq) * LDP + 16 * k + lc], acc);
#pragma unroll
                for (int g = 0; g < 4; ++g) sP[(16 * j + 4 * g + lq) * LDP + 16 * k + lc] = acc[g];
            }
            __syncthreads();
            {
                const int i = wave >> 2, k = wave & 3;
                if (i < bw && k < bw && i != j && k != j) {
                    d4 acc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = sP[(16 * i + 4 * g + lq) * LDP + 16 * k + lc];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc = MFMA_NEGA(sP[(16 * i + lc) * LDP + 16 * j + 4 * t + lq], sP[(16 * j + 4 * t + lq) * LDP + 16 * k + lc], acc);
#pragma unroll
                    for (int g = 0; g < 4; ++g) sP[(16 * i + 4 * g + lq) * LDP + 16 * k + lc] = acc[g];
                }
            }
            __syncthreads();
            if (wave < bw) {
                const int i = wave;
                if (i != j) {
                    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc = MFMA_NEGA(sP[(16 * i + lc) * LDP + 16 * j + 4 * t + lq], sD[(4 * t + lq) * LDD + lc], acc);
#pragma unroll
                    for (int g = 0; g < 4; ++g) sP[(16 * i + 4 * g + lq) * LDP + 16 * j + lc] = acc[g];
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) sP[(16 * j + 4 * g + lq) * LDP + 16 * j + lc] = sD[(4 * g + lq) * LDD + lc];
                }
            }
            __syncthreads();
        }
        STAMP(2);   // inversion: the tile steps
        // ---- U' = (U - E) P^-1, this wave's 16 x wd piece, as the TRANSPOSED product P^-T U': its accumulators are the A
        // operands of the update (ua[4 J + g]: row lc, k-block 4 J + g).  B operand of the product: u[t] = T[16 I + lc][p0 + 4 t + lq]
        double ua[16];
        {
            double u[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                u[t] = 0.0;
                if (owner && t < 4 * bw) {
                    if constexpr (FIRST) u[t] = Q_[(4 * t + lq) * n + 16 * I + lc];
                    else u[t] = tile(I, 4 * kb + (t >> 2))[(lc >> 2) * 64 + (lc & 3) * 16 + 4 * (t & 3) + lq];
                }
            }
            if constexpr (FIRST) {
#pragma unroll
                for (int t = 0; t < 16; ++t) mabs = max_abs(mabs, u[t]);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) if (16 * I + lc == 64 * kb + 4 * t + lq) u[t] -= 1.0;      // pivot rows carry P - I
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                if (J < bw) {
                    if (bw == 4) {
#pragma unroll
                        for (int t = 0; t < 16; ++t) acc = MFMA(sP[(4 * t + lq) * LDP + 16 * J + lc], u[t], acc);
                    } else {
#pragma unroll
                        for (int t = 0; t < 12; ++t)        // (a narrow last block)
                            if (t < 4 * bw) acc = MFMA(sP[(4 * t + lq) * LDP + 16 * J + lc], u[t], acc);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) ua[4 * J + g] = acc[g];
            }
        }
        STAMP(3);   // U'
        // ---- T <- T - U' V on the live column tiles, MC at a time (their pivot rows are in LDS): every wave walks its own row
        // tile with the next tile's loads in flight behind the sixteen MFMAs of the current one -- no barrier inside a mega-chunk
        for (int J0 = Jlo; J0 < nct; J0 += MC) {
            const int cnt = (nct - J0) < MC ? (nct - J0) : MC;
            if (J0 > Jlo) {
                __syncthreads();                            // everybody is done with the previous pivot rows
                stage(J0, cnt);
                __syncthreads();
            }
            STAMP(4);   // (diagnostic) staging of a later mega-chunk
            if (owner) {
                d4 ct;
#pragma unroll
                for (int g = 0; g < 4; ++g) ct[g] = FIRST ? t0(I, J0, g) : tile(I, J0)[g * 64 + lane];
#pragma unroll 1
                for (int jj = 0; jj < cnt; ++jj) {
                    d4 nx = {0.0, 0.0, 0.0, 0.0};
                    if (jj + 1 < cnt) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) nx[g] = FIRST ? t0(I, J0 + jj + 1, g) : tile(I, J0 + jj + 1)[g * 64 + lane];
                    }
                    if constexpr (FIRST) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) mabs = max_abs(mabs, ct[g]);
                    }
                    const double *const vb = sV + (jj << 10) + lane;
                    if (bw == 4) {
#pragma unroll
                        for (int s = 0; s < 16; ++s) ct = MFMA_NEGA(ua[s], vb[((s >> 2) << 8) + (s & 3) * 64], ct);
                    } else {
#pragma unroll
                        for (int s = 0; s < 12; ++s)
                            if (s < 4 * bw) ct = MFMA_NEGA(ua[s], vb[((s >> 2) << 8) + (s & 3) * 64], ct);
                    }
                    double *const dst = tile(I, J0 + jj);
#pragma unroll
                    for (int g = 0; g < 4; ++g) dst[g * 64 + lane] = ct[g];
                    ct = nx;
                }
            }
            STAMP(6);   // (diagnostic) this wave's tiles of the mega-chunk
        }
        return true;
    };

    const int kb_lo = MODE == 2 ? 1 : 0, kb_hi = MODE == 1 ? 1 : (nrt + 3) / 4;
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        const bool ok = pass(std::integral_constant<bool, RECORDS>{}, kb);
        if (!ok) { if (tid == 0) a.status[b] = -1; return; }
        if constexpr (RECORDS) {
            // max |M| is known now: the threshold of the remaining passes (left in the workspace for them), and the pivots of this
            // pass once more
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mabs, off, 64); mabs = o > mabs ? o : mabs; }
            if (lane == 0) sRed[wave] = mabs;
            if (tid == 0) sRed[16] = minpiv;            // (wave 0 ran the inversions)
            __syncthreads();
            double ms = sRed[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) ms = sRed[k] > ms ? sRed[k] : ms;
            thr = 1e-4 * (ms > 1.0 ? ms : 1.0);
            if (sRed[16] < thr) { if (tid == 0) a.status[b] = -1; return; }
            if (tid == 0) w.c[(size_t)b * (n + m)] = ms;
        }
        __threadfence_block();
        __syncthreads();
        STAMP(7);   // (diagnostic) end-of-pass barrier
    }
#ifdef QPN_STAMPS
    if (tid == 0 && a.stamps) for (int k = 0; k < 8; ++k) a.stamps[(size_t)b * 8 + k] += stamp_acc[k];
#endif
    // eliminated (-3: the S product takes it from here) / pass 0 done (-4: the remaining passes are another launch)
    if (tid == 0) a.status[b] = (MODE == 1 && kb_hi < (nrt + 3) / 4) ? -4 : -3;
}

// ---- K2: S = -Ad W and c = b - Ad h (b = B w): wave Is holds row tile Is of S, W | h crosses LDS in blocks of 64 rows x JC
// column tiles (plain copies of tiles: the B operands as they stand); then the reduced problem's bounds and cold start
__global__ __launch_bounds__(TPB2, 1) void schur_big2_sprod(AviBatchArgs a, SchurBigWs w)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    if (a.status[b] != -3) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    const int nrt = pad16(n) >> 4, mct = pad16(m) >> 4, nct = nrt + mct + 1;
    extern __shared__ __attribute__((aligned(32))) double sm[];
    double *const sV = sm, *const sBv = sm + 2 * 16 * 256;
    const double *A_ = a.nd.Ad + (size_t)b * m * n;
    const double *Tt = w.Tt + (size_t)b * (size_t)w.tt_stride;
    const size_t vo = (size_t)b * (size_t)N;
    double *Sg = w.S + (size_t)b * (size_t)w.s_stride;
    double *cg = w.c + vo;
    if (tid < m) {
        const double *B_ = a.nd.B + (size_t)b * m * np_;
        const double *w_ = a.nd.w + (size_t)b * (size_t)a.nd.stride_w;
        double s = 0.0;
        for (int k = 0; k < np_; ++k) s = fma(B_[(size_t)k * m + tid], w_[k], s);
        sBv[tid] = s;
    }
    const int Is = wave;
    const bool sown = Is < mct;
    const int nkc = (nrt + 3) / 4;                  // 64-row blocks of W
    const int ncc = (mct + 1 + JC - 1) / JC;        // chunks of column tiles (the last tile is h)
    constexpr int CHS = 16 * 256;                   // doubles per buffer: 4 block rows x JC tiles
    // staging: thread -> 32 bytes of tile (block row sti >> 2, column tile sti & 3); one buffer ahead, one barrier per step
    const int sti = tid >> 6, sbr = sti >> 2, sjj = sti & 3, se = (tid & 63) * 4;
    auto wl = [&](int it) -> d4 {
        const int cc = it / nkc, kc = it - cc * nkc;
        const int J = JC * cc + sjj, Ik = 4 * kc + sbr;
        d4 v = {0.0, 0.0, 0.0, 0.0};
        if (J <= mct && Ik < nrt) v = *reinterpret_cast<const d4 *>(Tt + ((size_t)(Ik * nct + nrt + J) << 8) + se);
        return v;
    };
    auto ws_ = [&](int it, d4 v) { *reinterpret_cast<d4 *>(sV + (it & 1) * CHS + sti * 256 + se) = v; };
    // A operands: -Ad[16 Is + lc][64 kc + 4 s + lq] (clamped addresses, the value selected afterwards)
    auto al = [&](int kc, double (&aop)[16]) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int ri = 16 * Is + lc, k = 64 * kc + 4 * s + lq;
            const bool ok = sown && ri < m && k < n;
            const double v = A_[ok ? k * m + ri : 0];
            aop[s] = ok ? -v : 0.0;
        }
    };
    ws_(0, wl(0));
    __syncthreads();
    d4 acc[JC];
    const int nit = ncc * nkc;
    for (int it = 0; it < nit; ++it) {
        const int cc = it / nkc, kc = it - cc * nkc;
        const bool more = it + 1 < nit;
        d4 vn = {0.0, 0.0, 0.0, 0.0};
        if (more) vn = wl(it + 1);
        if (sown) {
            if (kc == 0) {
#pragma unroll
                for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ri = 16 * Is + 4 * g + lq;
                        const double bv = sBv[ri < m ? ri : 0];
                        acc[jj][g] = (JC * cc + jj == mct && lc == 0 && ri < m) ? bv : 0.0;
                    }
            }
            double aop[16];
            al(kc, aop);
            const double *const vb = sV + (it & 1) * CHS;
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                if (JC * cc + jj <= mct) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[jj] = MFMA(aop[s], vb[((s >> 2) * 4 + jj) * 256 + (s & 3) * 64 + lane], acc[jj]);
                }
            }
            if (kc == nkc - 1) {
#pragma unroll
                for (int jj = 0; jj < JC; ++jj) {
                    const int J = JC * cc + jj;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ri = 16 * Is + 4 * g + lq, cj = 16 * J + lc;
                        if (J < mct) { if (ri < m && cj < m) Sg[(size_t)ri * ((m + 1 + 15) & ~15) + cj] = acc[jj][g]; }       // straight into the Lemke dictionary
                        else if (J == mct && lc == 0 && ri < m) cg[ri] = acc[jj][g];
                    }
                }
            }
        }
        if (more) ws_(it + 1, vn);
        __syncthreads();
    }
    // reduced problem data: bounds of the constraint rows, cold start
    for (int k = tid; k < m; k += TPB2) {
        w.l2[vo + k] = a.nd.l[(size_t)b * m + k]; w.u2[vo + k] = a.nd.u[(size_t)b * m + k]; w.lam[vo + k] = 0.0;
    }
    if (tid == 0) { w.nsplit[b] = n; w.nred[b] = m; a.status[b] = -2; }
}

// ---- finish: x = -(W lambda + h), post-check on the RECORDS ------------------------------------------------------------
constexpr int TPBF = 256;
__global__ __launch_bounds__(TPBF) void schur_big2_finish(AviBatchArgs a, SchurBigWs w)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    if (a.status[b] != -2) return;
    const int n = a.nd.n, m = a.nd.m, np_ = a.nd.p, N = n + m;
    __shared__ double red[TPBF / 64];
    __shared__ int redi[TPBF / 64];
    extern __shared__ __attribute__((aligned(16))) double zs[];       // z, N doubles
    const int n_pad = pad16(n), m_pad = pad16(m);
    const double *Tt = w.Tt + (size_t)b * (size_t)w.tt_stride;
    const double *Q_ = a.nd.Qd + (size_t)b * n * n;
    const double *A_ = a.nd.Ad + (size_t)b * m * n;
    const double *R_ = a.nd.R + (size_t)b * n * np_;
    const double *B_ = a.nd.B + (size_t)b * m * np_;
    const double *w_ = a.nd.w + (size_t)b * (size_t)a.nd.stride_w;
    const size_t vo = (size_t)b * (size_t)N;
    for (int k = tid; k < m; k += TPBF) zs[n + k] = w.lam[vo + k];
    __syncthreads();
    {
        // W | h in the tile layout of stage A: element (i, k) is register (i % 16) / 4 of lane 16 (i % 4) + k % 16 in tile
        // (i / 16, nrt + k / 16)
        const int nrt = n_pad >> 4, mct = m_pad >> 4, nct = nrt + mct + 1;
        for (int i = tid; i < n; i += TPBF) {
            const double *row = Tt + ((size_t)((i >> 4) * nct + nrt) << 8) + ((i & 15) >> 2) * 64 + (i & 3) * 16;
            double s = row[(size_t)mct << 8];
            for (int k = 0; k < m; ++k) s = fma(row[((size_t)(k >> 4) << 8) + (k & 15)], zs[n + k], s);
            zs[i] = -s;
        }
    }
    __syncthreads();
    int bad = 0;
    double nres = 0.0;
    for (int k = tid; k < N; k += TPBF) {
        // r = q + M z, item columns in ascending order (finite blocks: a zero z_j contributes exactly nothing)
        const bool isx = k < n;
        const int ks = isx ? k : k - n;
        const double *col = isx ? R_ + ks : B_ + ks;
        const size_t cs = isx ? (size_t)n : (size_t)m;
        double rk = isx ? a.nd.qd[(size_t)b * n + ks] : 0.0;
        for (int t = 0; t < np_; ++t) rk = fma(col[(size_t)t * cs], w_[t], rk);
        if (isx) {
            int j = 0;
            for (; j + 8 <= n; j += 8) {                   // Qd column-wise: lane <-> row, coalesced
                double mv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) mv[q8] = Q_[(size_t)(j + q8) * n + ks];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) rk = fma(mv[q8], zs[j + q8], rk);
            }
            for (; j < n; ++j) rk = fma(Q_[(size_t)j * n + ks], zs[j], rk);
            const double *ar = A_ + (size_t)ks * m;        // -Ad' row: column ks of Ad (contiguous)
            j = 0;
            for (; j + 8 <= m; j += 8) {
                double mv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) mv[q8] = ar[j + q8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) rk = fma(-mv[q8], zs[n + j + q8], rk);
            }
            for (; j < m; ++j) rk = fma(-ar[j], zs[n + j], rk);
        } else {
            int j = 0;
            for (; j + 8 <= n; j += 8) {                   // Ad column-wise: lane <-> constraint row, coalesced
                double mv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) mv[q8] = A_[(size_t)(j + q8) * m + ks];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) rk = fma(mv[q8], zs[j + q8], rk);
            }
            for (; j < n; ++j) rk = fma(A_[(size_t)j * m + ks], zs[j], rk);
        }
        const int gk = !isx;
        const double zk = zs[k];
        const double lk = isx ? -QINF : a.nd.l[(size_t)b * m + ks], uk = isx ? QINF : a.nd.u[(size_t)b * m + ks];
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
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_xor(bad, off, 64); const double o = __shfl_xor(nres, off, 64); nres = o > nres ? o : nres; }
    if ((tid & 63) == 0) { red[tid >> 6] = nres; redi[tid >> 6] = bad; }
    __syncthreads();
    if (tid == 0) {
        int badt = 0; double nrest = 0.0;
        for (int k = 0; k < TPBF / 64; ++k) { badt += redi[k]; nrest = red[k] > nrest ? red[k] : nrest; }
        int status = w.st2[b];
        if (badt > 0 && status == QPN_SUCCESS) status = QPN_FAILURE;
        a.status[b] = status;
        if (a.resid) a.resid[b] = nrest;
        if (a.pivots) a.pivots[b] = n + w.piv2[b];
    }
}

} // namespace

bool qpn_schur_big2_shape(int n, int m)
{
    return n > 64 && n <= 256 && m >= 1 && m <= 256;
}

// Stage A from the records (a.nd set, a.N = n + m); `ws` is carved exactly as qpn_launch_schur_big_stage_a carves it (S row-major:
// the delayed-update Lemke kernel behind it).
hipError_t qpn_launch_schur_big2_stage_a(const AviBatchArgs &a, void *ws, double *dict, SchurBigWs *out, hipStream_t stream)
{
    const int N = a.N, batch = a.batch;
    const size_t rows = (size_t)((N + 15) & ~15);
    SchurBigWs w{};
    w.tt_stride = (int64_t)((rows < 512 ? rows : 512) * (size_t)(((N + 15) & ~15) + 48));
    w.s_stride = (int64_t)N * N;
    w.s_rowmajor = 2;                                   // S is written straight into the Lemke kernel's dictionary (its row stride)
    double *p = static_cast<double *>(ws);
    w.Tt = p; p += (size_t)batch * w.tt_stride;
    p += (size_t)batch * w.s_stride;                    // (the S area of the shared carving: unused here)
    w.c = p; p += (size_t)batch * N;
    w.l2 = p; p += (size_t)batch * N;
    w.u2 = p; p += (size_t)batch * N;
    w.lam = p; p += (size_t)batch * N;
    int32_t *ip = reinterpret_cast<int32_t *>(p);
    w.st2 = ip; ip += batch; w.piv2 = ip; ip += batch; w.nsplit = ip; ip += batch; w.nred = ip; ip += batch;
    w.S = dict; w.s_stride = (int64_t)N * (N + 1);      // item b's dictionary: dict + b N (N + 1), as the Lemke kernel addresses it
    *out = w;
    // (the top half of a node: pad16(n) rows of pad16(n) + pad16(m) + 16 doubles -- inside the stride above for n, m <= 256)
    const size_t need = (size_t)((a.nd.n + 15) & ~15) * (size_t)(((a.nd.n + 15) & ~15) + ((a.nd.m + 15) & ~15) + 16);
    if (!qpn_schur_big2_shape(a.nd.n, a.nd.m) || a.nd.n + a.nd.m != N || need > (size_t)w.tt_stride) return hipErrorInvalidValue;
    static QpnPerDeviceOnce attr_once;
    const int attr_dev = attr_once.device();
    if (!attr_once.done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big2_eliminate<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DOUBLES * (int)sizeof(double));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big2_eliminate<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DOUBLES * (int)sizeof(double));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big2_eliminate<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DOUBLES * (int)sizeof(double));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_big2_sprod),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_SPROD * (int)sizeof(double));
        if (e != hipSuccess) return e;
        attr_once.done[attr_dev] = true;
    }
    const dim3 grid((unsigned)batch);
    static const bool force_convert = [] { const char *e = QPN_DEV_ENV("QPN_BIG2_CONVERT"); return e && e[0] == '1'; }();     // A/B switch
    if (!force_convert && (a.nd.n & 15) == 0 && (a.nd.m & 15) == 0) {
        hipLaunchKernelGGL(schur_big2_eliminate<1>, grid, dim3(TPB2), LDS_DOUBLES * sizeof(double), stream, a, w);
        if (a.nd.n > 64) hipLaunchKernelGGL(schur_big2_eliminate<2>, grid, dim3(TPB2), LDS_DOUBLES * sizeof(double), stream, a, w);
    } else {
        hipLaunchKernelGGL(schur_big2_convert, grid, dim3(256), 0, stream, a, w);
        hipLaunchKernelGGL(schur_big2_eliminate<0>, grid, dim3(TPB2), LDS_DOUBLES * sizeof(double), stream, a, w);
    }
    hipLaunchKernelGGL(schur_big2_sprod, grid, dim3(TPB2), LDS_SPROD * sizeof(double), stream, a, w);
    return hipGetLastError();
}

hipError_t qpn_launch_schur_big2_finish(const AviBatchArgs &a, const SchurBigWs &w, hipStream_t stream)
{
    hipLaunchKernelGGL(schur_big2_finish, dim3((unsigned)a.batch), dim3(TPBF), (size_t)a.N * sizeof(double), stream, a, w);
    return hipGetLastError();
}
