// qpn_avi_reg.hip -- register-resident batched box-MCP / GAVI pivotal solver for gfx950 (CDNA4).
//
// Same algorithm and the same arithmetic (bit for bit) as qpn_avi_solve.hip / the CPU oracle;
// different machine mapping.  ONE 64-lane wavefront per AVI; the 64 lanes form an 8 x 8 grid and
// lane (ra, cb) keeps the BS x BS block  T[BS*ra .. , BS*cb ..]  of the dictionary in VGPRs
// (BS = 8 for N <= 64: 128 VGPRs).  A pivot is a rank-1 update  T -= u v'  in which every lane
// needs only BS entries of u (pivot column) and BS entries of v (scaled pivot row): both
// vectors go through two tiny padded LDS arrays (conflict-free ds_read_b128), so a pivot costs
// BS*BS v_fma_f64 per lane and ~16 LDS instructions instead of streaming the whole tableau
// through LDS.  Dynamic row/column selection never indexes registers dynamically: the pivot
// column / row are copied out by the 8 lanes that own them inside a wave-uniform switch.
// Row- and column-vectors (basic values, covering column, bookkeeping) live one element per lane.
// HBM traffic: the stacked M block is read with coalesced 512-byte column loads, staged BS
// columns at a time through LDS; the post-check (src/avi.jl:71-76) re-reads it (L2/MALL).
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

template <int BS> struct Geo {
    static constexpr int NB = 8 * BS;        // padded dimension held in registers
    static constexpr int PB = BS + 2;        // padded block stride (doubles) in the LDS vectors
    static constexpr int NP = 8 * PB;        // padded vector length
    __device__ static __forceinline__ int pidx(int i) { return (i / BS) * PB + (i % BS); }
};

__device__ __forceinline__ void var_interval_r(int v, int N, const double *sl, const double *su,
                                               const int *sat, double &lo, double &hi)
{
    if (v < 0) { lo = -QINF; hi = QINF; return; }
    if (v == 2 * N) { lo = 0.0; hi = QINF; return; }
    if (v < N) { lo = sl[v]; hi = su[v]; return; }
    int k = v - N;
    double L = sl[k], U = su[k];
    if (L == U) { lo = -QINF; hi = QINF; }
    else if (L == -QINF && U == QINF) { lo = 0.0; hi = 0.0; }
    else if (sat[k]) { lo = -QINF; hi = 0.0; }
    else { lo = 0.0; hi = QINF; }
}

template <int BS>
__global__ __launch_bounds__(WAVE, 2) void avi_solve_reg(AviBatchArgs a)
{
    using G = Geo<BS>;
    constexpr int NB = G::NB, PB = G::PB, NP = G::NP;
    constexpr int XC = NB;                 // index of the extra (covering) column
    const int N = a.N;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (a.only_if && a.only_if[b] != a.only_if_value) return;   // wave-uniform gate
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
    const size_t vo = (size_t)b * (size_t)N;
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;

    // ---- pair k = lane ----------------------------------------------------------------
    const double lk = act ? a.l[vo + lane] : 0.0;
    const double uk = act ? a.u[vo + lane] : 0.0;
    const int gk = (act && a.kind) ? (int)a.kind[(size_t)b * (size_t)a.stride_kind + lane] : 0;
    const bool freek = act && lk == -QINF && uk == QINF;
    const bool fixedk = act && lk == uk;
    int atup0 = 0;
    double v0 = 0.0;
    int rowvar = -1, colvar = -1;
    if (act) {
        if (gk) { colvar = N + lane; rowvar = lane; }
        else {
            double z0 = a.z[vo + lane];
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
    double nbval = v0;
    int cNvar = 2 * N;
    double cNval = 0.0;
    double tcol = 0.0;                      // extra column, one entry per row-lane
    double xb = act ? a.q[vo + lane] : 0.0;
    __syncthreads();

    // ---- load M: coalesced HBM -> LDS stage (BS columns at a time) -> register blocks ------
    double t[BS][BS];
#pragma unroll
    for (int k = 0; k < BS; ++k)
#pragma unroll
        for (int l = 0; l < BS; ++l) t[k][l] = 0.0;
    for (int cbk = 0; cbk < 8; ++cbk) {
        const int col0 = BS * cbk;
        if (col0 >= N) break;
        const int ncols = (N - col0) < BS ? (N - col0) : BS;
        const int cnt = ncols * N;
        const double *base = Mg + (size_t)col0 * N;
        {
            int row = lane, col = 0;
            while (row >= N) { row -= N; col++; }
            for (int idx = lane; idx < cnt; idx += WAVE) {
                stage[col * NP + G::pidx(row)] = base[idx];
                row += WAVE;
                while (row >= N) { row -= N; col++; }
            }
        }
        __syncthreads();
        // initial basic values  xb = q + M z_nb, columns in ascending order (as the oracle)
        for (int l = 0; l < ncols; ++l) {
            const double zj = snb[col0 + l];
            if (zj != 0.0 && act) xb = fma(stage[l * NP + G::pidx(lane)], zj, xb);
        }
        if (cb == cbk) {
#pragma unroll
            for (int l = 0; l < BS; ++l)
#pragma unroll
                for (int k = 0; k < BS; ++k) {
                    const int row = BS * ra + k;
                    t[k][l] = (row < N && l < ncols) ? stage[l * NP + ra * PB + k] : 0.0;
                }
        }
        __syncthreads();
    }

    STAMP(0);   // setup + load
    int pivots = 0;

    // Dynamic row / column selection without dynamic register indexing.  The selector becomes a
    // one-hot mask tested bit by bit with wave-uniform branches; the branch bodies are either LDS
    // stores or `asm volatile` register writes, which hipcc can neither speculate nor turn into
    // selects, so every body touches statically named registers IN PLACE.  (A `switch`, or plain
    // C++ assignments, make hipcc compute all BS*BS candidates and select -- hundreds of moves.)
    auto extract_col = [&](int c_in) -> double {
        const int c = uni(c_in);
        if (c == XC) {
            if (lane < NB) ucol[G::pidx(lane)] = tcol;
        } else {
            const unsigned hot = 1u << (c % BS);
            if (cb == c / BS) {
#pragma unroll
                for (int L = 0; L < BS; ++L) {
                    if (hot & (1u << L)) {
#pragma unroll
                        for (int k = 0; k < BS; ++k) ucol[ra * PB + k] = t[k][L];
                    }
                }
            }
        }
        __syncthreads();
        return lane < NB ? ucol[G::pidx(lane)] : 0.0;
    };

    // exchange the entering variable (column c, moved by delta) with the basic variable of row r;
    // cm = this row-lane's entry of column c (ucol still holds the column)
    auto do_pivot = [&](int r_in, int c_in, double delta, double leave_val, double cm) {
        const int r = uni(r_in), c = uni(c_in);
        xb = fma(delta, cm, xb);
        const double enter_old = (c == XC) ? cNval : readlane_f64(nbval, c);
        const double enter_val = enter_old + delta;
        const double inv = 1.0 / readlane_f64(cm, r);
        const unsigned rhot = 1u << (r % BS);
        const bool rmine = ra == r / BS;
        const unsigned chot = (c == XC) ? 0u : (1u << (c % BS));
        const bool cmine = (c != XC) && cb == c / BS;
        // raw pivot row -> vrow (scaled by the readers)
        if (rmine) {
#pragma unroll
            for (int K = 0; K < BS; ++K) {
                if (rhot & (1u << K)) {
#pragma unroll
                    for (int l = 0; l < BS; ++l) vrow[cb * PB + l] = t[K][l];
                }
            }
        }
        if (lane == r) vrow[NP] = tcol;
        __syncthreads();
        double u[BS], v[BS];
#pragma unroll
        for (int k = 0; k < BS; ++k) u[k] = ucol[ra * PB + k];
#pragma unroll
        for (int l = 0; l < BS; ++l) {
            const double raw = vrow[cb * PB + l];
            // prow_j = T[r][j] * inv; the slot of column c carries -inv (row fix-up below)
            v[l] = (BS * cb + l == c) ? -inv : raw * inv;
        }
        const double vx = (c == XC) ? -inv : vrow[NP] * inv;
#pragma unroll
        for (int k = 0; k < BS; ++k)
#pragma unroll
            for (int l = 0; l < BS; ++l) t[k][l] = fma(-u[k], v[l], t[k][l]);
        // column c of the new dictionary: T[i][c] = cm_i * inv
        if (cmine) {
#pragma unroll
            for (int L = 0; L < BS; ++L) {
                if (chot & (1u << L)) {
#pragma unroll
                    for (int k = 0; k < BS; ++k)
                        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t[k][L]) : "v"(u[k]), "v"(inv));
                }
            }
        }
        // row r of the new dictionary: T[r][j] = -prow_j, T[r][c] = inv (v holds -inv there)
        if (rmine) {
#pragma unroll
            for (int K = 0; K < BS; ++K) {
                if (rhot & (1u << K)) {
#pragma unroll
                    for (int l = 0; l < BS; ++l)
                        asm volatile("v_mul_f64 %0, %1, -1.0" : "=v"(t[K][l]) : "v"(v[l]));
                }
            }
        }
        // extra column
        if (c == XC) tcol = (lane == r) ? inv : cm * inv;
        else tcol = (lane == r) ? -vx : fma(-cm, vx, tcol);
        // bookkeeping
        const int ve = (c == XC) ? cNvar : readlane_i32(colvar, c);
        const int vl = readlane_i32(rowvar, r);
        if (lane == r) { rowvar = ve; xb = enter_val; }
        if (c == XC) { cNvar = vl; cNval = leave_val; }
        else if (lane == c) { colvar = vl; nbval = leave_val; }
        __syncthreads();
    };
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
    __syncthreads();
    int budget = 4 * N + 4;
    int stage_ = 0, idx = 0;
    int status = QPN_FAILURE;
    int c = XC;
    double sigma = -1.0, self_lim = 0.0;
    const double slack = 1e-10;
    const double ptol = a.piv_tol;
    for (;;) {
        if (stage_ == 0) {
            if (!(idx < n_enter && budget > 0)) { stage_ = 1; continue; }
            const int e = uni(elist[idx]);
            idx++;
            c = wave_first(act && colvar == e);
            if (c < 0) continue;
        } else if (stage_ == 1) {
            double lo, hi;
            var_interval_r(rowvar, N, sl, su, sat, lo, hi);
            double viol = 0.0;
            if (act) viol = xb < lo ? lo - xb : (xb > hi ? xb - hi : 0.0);
            const double theta0 = wave_max_f64(viol);
            if (theta0 <= a.feas_tol) { status = QPN_SUCCESS; break; }
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
            status = QPN_MAX_ITERS;
            stage_ = 2;
            continue;
        } else {
            if (pivots >= max_piv) break;
        }

        STAMP(1);   // loop control / stage setup
        const double cm = extract_col(c);
        STAMP(2);   // column extraction
        int r;
        double delta, leave_val;
        if (stage_ == 0) {
            const double av = act ? fabs(cm) : 0.0;
            const double colmax = wave_max_f64(av);
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
            const double thresh = 1e-9 * (colmax > 1.0 ? colmax : 1.0);
            double best = wave_max_f64(ml ? av : -1.0);
            if (best > thresh) {
                r = wave_first(ml && av == best);
                leave_val = readlane_f64(tg, r);
            } else {
                // no equation row can take it: 2x2 principal block pivot through an ordinary pair
                best = wave_max_f64(ord ? av : -1.0);
                if (!(best > thresh)) continue;
                r = wave_first(ord && av == best);
                const int v = readlane_i32(rowvar, r);
                if (v < N) {
                    const double x = readlane_f64(xb, r), lo = sl[v], hi = su[v];
                    int au;
                    if (x <= lo) { leave_val = lo; au = 0; }
                    else if (x >= hi) { leave_val = hi; au = 1; }
                    else if (lo == -QINF) { leave_val = hi; au = 1; }
                    else if (hi == QINF) { leave_val = lo; au = 0; }
                    else if (hi - x < x - lo) { leave_val = hi; au = 1; }
                    else { leave_val = lo; au = 0; }
                    if (lane == 0) { sat[v] = au; elist[n_enter] = N + v; }
                } else {
                    leave_val = 0.0;
                    if (lane == 0) elist[n_enter] = v - N;
                }
                n_enter++;
            }
            delta = (leave_val - readlane_f64(xb, r)) / readlane_f64(cm, r);
        } else {
            double lo, hi;
            var_interval_r(rowvar, N, sl, su, sat, lo, hi);
            const double g = act ? sigma * cm : 0.0;
            double d1 = QINF;
            bool cnd = false;
            double d = 0.0, lv = 0.0;
            if (act) {
                if (g < -ptol && lo > -QINF) { d1 = (xb - lo + slack) / (-g); d = (xb - lo) / (-g); lv = lo; cnd = true; }
                else if (g > ptol && hi < QINF) { d1 = (hi - xb + slack) / g; d = (hi - xb) / g; lv = hi; cnd = true; }
            }
            double dmax = wave_min_f64(d1);
            if (self_lim < dmax) dmax = self_lim;
            if (dmax == QINF) { status = QPN_RAY_TERM; break; }
            if (cnd && d > dmax) cnd = false;
            double ag = cnd ? fabs(g) : -1.0;
            if (cnd && rowvar == 2 * N) ag = QINF;
            const double bestg = wave_max_f64(ag);
            if (bestg < 0.0) {
                // the entering variable reaches its own far bound first
                const double dl = sigma * self_lim;
                if (act) xb = fma(dl, cm, xb);
                const int ve = (c == XC) ? cNvar : readlane_i32(colvar, c);
                if (ve == 2 * N) {
                    if (c == XC) cNval = 0.0; else if (lane == c) nbval = 0.0;
                    status = QPN_SUCCESS; break;
                }
                const int k = ve;
                const int au = sigma > 0.0 ? 1 : 0;
                const double nv = au ? su[k] : sl[k];
                if (lane == 0) sat[k] = au;
                if (c == XC) cNval = nv; else if (lane == c) nbval = nv;
                pivots++;
                c = col_of(N + k);
                if (c < 0) { status = QPN_FAILURE; break; }
                sigma = au ? -1.0 : 1.0;
                self_lim = QINF;
                __syncthreads();
                continue;
            }
            r = wave_first(cnd && ag == bestg);
            double step = readlane_f64(d, r);
            if (step < 0.0) step = 0.0;
            leave_val = readlane_f64(lv, r);
            delta = sigma * step;
        }

        const int vl = readlane_i32(rowvar, r);
        STAMP(3);   // pivot selection (ratio tests, reductions)
        do_pivot(r, c, delta, leave_val, cm);
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
                const double Lk = sl[k], Uk = su[k];
                int au = sat[k];
                if (Lk != Uk) { au = (leave_val == Uk) ? 1 : 0; if (lane == 0) sat[k] = au; }
                vn = N + k;
                sigma = au ? -1.0 : 1.0;
                self_lim = QINF;
            } else {
                const int k = vl - N;
                const double Lk = sl[k], Uk = su[k];
                vn = k;
                sigma = sat[k] ? -1.0 : 1.0;
                self_lim = Uk - Lk;
                if (Lk == -QINF && Uk == QINF) { self_lim = QINF; sigma = 1.0; }
            }
            c = col_of(vn);
            if (c < 0) { status = QPN_FAILURE; break; }
            __syncthreads();
        }
    }

    STAMP(1);
    // ---- read the point back ----------------------------------------------------------------
    __syncthreads();
    if (act) { snb[rowvar] = xb; snb[colvar] = nbval; }
    if (lane == 0) snb[cNvar] = cNval;
    __syncthreads();
    const double zk = act ? snb[gk ? N + lane : lane] : 0.0;
    __syncthreads();
    if (act) ucol[lane] = zk;   // z, broadcast source for the post-check mat-vec (NB <= NP)
    __syncthreads();

    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -------------------
    double rk = act ? a.q[vo + lane] : 0.0;
    for (int j = 0; j < N; ++j) {
        double zj = ucol[j];
        if (zj != 0.0 && act) rk = fma(Mg[(size_t)j * N + lane], zj, rk);
    }
    const double p = gk ? rk : zk, d = gk ? zk : rk;
    int bad = 0;
    double nres = 0.0;
    unsigned mask = 0;
    if (act) {
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

} // namespace

hipError_t qpn_launch_avi_solve_reg(const AviBatchArgs &a, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    const dim3 grid((unsigned)a.batch), block(WAVE);
    if (a.N <= 8) hipLaunchKernelGGL(avi_solve_reg<1>, grid, block, 0, stream, a);
    else if (a.N <= 16) hipLaunchKernelGGL(avi_solve_reg<2>, grid, block, 0, stream, a);
    else if (a.N <= 32) hipLaunchKernelGGL(avi_solve_reg<4>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(avi_solve_reg<8>, grid, block, 0, stream, a);
    return hipGetLastError();
}
