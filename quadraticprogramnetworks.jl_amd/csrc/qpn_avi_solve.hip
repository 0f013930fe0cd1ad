// qpn_avi_solve.hip -- batched box-MCP / GAVI pivotal solver for gfx950 (CDNA4).
//
// Replaces the PATHSolver.solve_mcp call of src/avi.jl:64-70 (+ post-check :71-76,
// check_avi_solution :148-156, comp_indices src/avi_solutions.jl:511-562) for batches of small
// independent node-AVIs (N <= 64).
//
// Kernel "lds1": ONE 64-lane wavefront per AVI.  Lane i owns row i of the dictionary
// (tableau) and pair i's bookkeeping; the N x (N+1) tableau lives in LDS column-major with
// an odd leading dimension, so a column access (lane i -> row i) is conflict-free ds_read_b64 /
// ds_write_b64 and the pivot-row gather is at most 2-way conflicted.  The stacked M/q blocks
// are read from HBM exactly twice (load, post-check; the second read is served by L2/MALL)
// with 512-byte coalesced column loads.  Pivot selection (Stage A partial pivoting, Stage B
// two-pass ratio test) uses wave-wide reductions (__shfl_xor / __ballot); every branch is
// wave-uniform.  Algorithm: see DESIGN.md section 3 ("pair dictionary, crash, Lemke").
#include <cstdlib>
#include <cstring>

#include "qpn_internal.h"

#define QINF __builtin_huge_val()

namespace {

constexpr int WAVE = 64;

struct LdsLayout {
    int LD, NC;
    size_t off_prow, off_sl, off_su, off_snb, off_sat, off_elist, bytes;
};

__host__ __device__ inline LdsLayout lds_layout(int N)
{
    LdsLayout L;
    L.NC = N + 1;
    L.LD = N | 1;
    size_t o = (size_t)L.LD * L.NC;          // T
    L.off_prow = o; o += (size_t)(L.NC + 1) & ~(size_t)1;
    L.off_sl = o;   o += (size_t)N;
    L.off_su = o;   o += (size_t)N;
    L.off_snb = o;  o += (size_t)(2 * N + 2); // nonbasic values at init, var-indexed values at read-back
    size_t ib = o * sizeof(double);
    L.off_sat = ib;   ib += sizeof(int) * (size_t)N;
    L.off_elist = ib; ib += sizeof(int) * (size_t)(8 * N + 8);
    L.bytes = (ib + 15) & ~(size_t)15;
    return L;
}

// admissible interval of variable v while it is basic (pair semantics: DESIGN.md section 3)
__device__ __forceinline__ void var_interval(int v, int N, const double *sl, const double *su,
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

__global__ __launch_bounds__(WAVE) void avi_solve_lds1(AviBatchArgs a)
{
    const int N = a.N;
    const LdsLayout L = lds_layout(N);
    const int NC = L.NC, LD = L.LD;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const bool act = lane < N;
    if (a.only_if && a.only_if[b] != a.only_if_value) return;  // wave-uniform gate

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *T = reinterpret_cast<double *>(smem_raw);
    double *prow = T + L.off_prow;
    double *sl = T + L.off_sl;
    double *su = T + L.off_su;
    double *snb = T + L.off_snb;
    int *sat = reinterpret_cast<int *>(smem_raw + L.off_sat);
    int *elist = reinterpret_cast<int *>(smem_raw + L.off_elist);

    const double *Mg = a.M + (size_t)b * (size_t)a.strideM;
    const size_t vo = (size_t)b * (size_t)N;
    const int max_piv = a.max_pivots > 0 ? a.max_pivots : 50 * N + 100;

    // ---- pair k = lane: bounds, kind, initial nonbasic value -------------------------
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
    double nbval = v0;
    int cNvar = 2 * N;       // column N: the artificial (wave-uniform bookkeeping)
    double cNval = 0.0;

    // ---- load the stacked M block: coalesced column loads, HBM -> LDS ------------------
    if (act) {
        for (int j = 0; j < N; ++j) T[j * LD + lane] = Mg[(size_t)j * N + lane];
        T[N * LD + lane] = 0.0;
    }
    __syncthreads();
    double xb = act ? a.q[vo + lane] : 0.0;
    for (int j = 0; j < N; ++j) {
        double zj = snb[j];
        if (zj != 0.0 && act) xb = fma(T[j * LD + lane], zj, xb);
    }

    int pivots = 0;
    double mabs = 0.0;
    if (act) for (int j = 0; j < N; ++j) mabs = fmax(mabs, fabs(T[j * LD + lane]));
    const double mscale = wave_max_f64(mabs);
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);

    // exchange the entering variable (column c, moved by delta) with the basic variable of row r
    auto do_pivot = [&](int r, int c, double delta, double leave_val) {
        double cm = act ? T[c * LD + lane] : 0.0;
        xb = fma(delta, cm, xb);
        double enter_old = (c == N) ? cNval : __shfl(nbval, c, WAVE);
        double enter_val = enter_old + delta;
        double inv = 1.0 / T[c * LD + r];
        if (act) prow[lane] = T[lane * LD + r] * inv;
        if (lane == 0) prow[N] = T[N * LD + r] * inv;
        __syncthreads();
        if (act) {
            const bool isr = lane == r;
            for (int j = 0; j < NC; ++j) {
                if (j == c) continue;
                double pr = prow[j];
                double t = T[j * LD + lane];
                T[j * LD + lane] = isr ? -pr : fma(-cm, pr, t);
            }
            T[c * LD + lane] = isr ? inv : cm * inv;
        }
        int ve = (c == N) ? cNvar : __shfl(colvar, c, WAVE);
        int vl = __shfl(rowvar, r, WAVE);
        if (lane == r) { rowvar = ve; xb = enter_val; }
        if (c == N) { cNvar = vl; cNval = leave_val; }
        else if (lane == c) { colvar = vl; nbval = leave_val; }
        __syncthreads();
    };
    auto col_of = [&](int v) -> int {
        int c = wave_first(act && colvar == v);
        if (c < 0 && cNvar == v) c = N;
        return c;
    };

    // ---- Stage A: crash.  Free variables / multipliers of equality GAVI rows enter -----
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
    for (int idx = 0; idx < n_enter && budget > 0; ++idx) {
        int e = uni(elist[idx]);
        int c = wave_first(act && colvar == e);
        if (c < 0) continue;
        double av = act ? fabs(T[c * LD + lane]) : 0.0;
        double colmax = wave_max_f64(av);
        bool ml = false, ord = false;
        double tg = 0.0;
        if (act) {
            int v = rowvar;
            if (v < 2 * N) {
                int k = v < N ? v : v - N;
                double Lk = sl[k], Uk = su[k];
                bool fr = Lk == -QINF && Uk == QINF, fx = Lk == Uk;
                if (v >= N) { if (fr) { ml = true; tg = 0.0; } }
                else if (fx && !fr) { ml = true; tg = Lk; }
                ord = !fr && !fx;
            }
        }
        double thresh = 1e-9 * (colmax > 1.0 ? colmax : 1.0);
        double best = wave_max_f64(ml ? av : -1.0);
        int rdiag = -1;
        if (e < N && c == e && __shfl(rowvar, e, WAVE) == N + e) {   // diagonal first
            const double ad = __shfl(av, e, WAVE);
            if (ad >= diag_thr) rdiag = e;
        }
        if (rdiag >= 0 || best > thresh) {
            int r = rdiag >= 0 ? rdiag : wave_first(ml && av == best);
            double target = rdiag >= 0 ? 0.0 : __shfl(tg, r, WAVE);
            double delta = (target - __shfl(xb, r, WAVE)) * (1.0 / T[c * LD + r]);
            do_pivot(r, c, delta, target);
            pivots++; budget--;
            continue;
        }
        // no equation row can take it: 2x2 principal block pivot through an ordinary pair
        best = wave_max_f64(ord ? av : -1.0);
        if (!(best > thresh)) continue;
        int r = wave_first(ord && av == best);
        int v = __shfl(rowvar, r, WAVE);
        double target;
        if (v < N) {
            double x = __shfl(xb, r, WAVE), lo = sl[v], hi = su[v];
            int au;
            if (x <= lo) { target = lo; au = 0; }
            else if (x >= hi) { target = hi; au = 1; }
            else if (lo == -QINF) { target = hi; au = 1; }
            else if (hi == QINF) { target = lo; au = 0; }
            else if (hi - x < x - lo) { target = hi; au = 1; }
            else { target = lo; au = 0; }
            if (lane == 0) { sat[v] = au; elist[n_enter] = N + v; }
        } else {
            target = 0.0;
            if (lane == 0) elist[n_enter] = v - N;
        }
        n_enter++;
        double delta = (target - __shfl(xb, r, WAVE)) * (1.0 / T[c * LD + r]);
        do_pivot(r, c, delta, target);  // ends with a barrier: elist/sat writes are visible
        pivots++; budget--;
        if (n_enter >= 8 * N) break;
    }

    // ---- Stage B: Lemke's complementary pivoting from the crash basis -------------------
    int status;
    {
        double lo, hi;
        var_interval(rowvar, N, sl, su, sat, lo, hi);
        double viol = 0.0;
        if (act) viol = xb < lo ? lo - xb : (xb > hi ? xb - hi : 0.0);
        const double theta0 = wave_max_f64(viol);
        if (theta0 <= a.feas_tol) {
            status = QPN_SUCCESS;
        } else {
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
                T[N * LD + lane] = cov;
            }
            cNval = theta0;
            int c = N;
            double sigma = -1.0, self_lim = theta0;
            status = QPN_MAX_ITERS;
            const double slack = 1e-10;
            const double ptol = a.piv_tol;
            __syncthreads();
            while (pivots < max_piv) {
                var_interval(rowvar, N, sl, su, sat, lo, hi);
                const double g = act ? sigma * T[c * LD + lane] : 0.0;
                // pass 1: step bound with slack
                double d1 = QINF;
                bool cnd = false;
                double d = 0.0, lv = 0.0;
                if (act) {
                    const double rc = 1.0 / g;
                    if (g < -ptol && lo > -QINF) { d = (xb - lo) * (-rc); d1 = d + slack * (-rc); lv = lo; cnd = true; }
                    else if (g > ptol && hi < QINF) { d = (hi - xb) * rc; d1 = d + slack * rc; lv = hi; cnd = true; }
                }
                double dmax = wave_min_f64(d1);
                if (self_lim < dmax) dmax = self_lim;
                if (dmax == QINF) { status = QPN_RAY_TERM; break; }
                // pass 2: largest pivot among the rows inside the bound; the artificial first
                if (cnd && d > dmax) cnd = false;
                double ag = cnd ? fabs(g) : -1.0;
                if (cnd && rowvar == 2 * N) ag = QINF;
                double bestg = wave_max_f64(ag);
                if (bestg < 0.0) {
                    // the entering variable reaches its own far bound first
                    double delta = sigma * self_lim;
                    if (act) xb = fma(delta, T[c * LD + lane], xb);
                    int ve = (c == N) ? cNvar : __shfl(colvar, c, WAVE);
                    if (ve == 2 * N) {
                        if (c == N) cNval = 0.0; else if (lane == c) nbval = 0.0;
                        status = QPN_SUCCESS; break;
                    }
                    int k = ve;
                    int au = sigma > 0.0 ? 1 : 0;
                    double nv = au ? su[k] : sl[k];
                    if (lane == 0) sat[k] = au;
                    if (c == N) cNval = nv; else if (lane == c) nbval = nv;
                    pivots++;
                    c = col_of(N + k);
                    if (c < 0) { status = QPN_FAILURE; break; }
                    sigma = au ? -1.0 : 1.0;
                    self_lim = QINF;
                    __syncthreads();
                    continue;
                }
                int r = wave_first(cnd && ag == bestg);
                double step = __shfl(d, r, WAVE);
                if (step < 0.0) step = 0.0;
                double leave_val = __shfl(lv, r, WAVE);
                int vl = __shfl(rowvar, r, WAVE);
                do_pivot(r, c, sigma * step, leave_val);
                pivots++;
                if (vl == 2 * N) { status = QPN_SUCCESS; break; }
                int vn;
                if (vl < N) {
                    int k = vl;
                    double Lk = sl[k], Uk = su[k];
                    int au = sat[k];
                    if (Lk != Uk) { au = (leave_val == Uk) ? 1 : 0; if (lane == 0) sat[k] = au; }
                    vn = N + k;
                    sigma = au ? -1.0 : 1.0;
                    self_lim = QINF;
                } else {
                    int k = vl - N;
                    double Lk = sl[k], Uk = su[k];
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
    }

    // ---- read the point back: scatter every variable's value by id, gather z_k ---------
    __syncthreads();
    if (act) { snb[rowvar] = xb; snb[colvar] = nbval; }
    if (lane == 0) snb[cNvar] = cNval;
    __syncthreads();
    double zk = act ? snb[gk ? N + lane : lane] : 0.0;
    __syncthreads();
    if (act) prow[lane] = zk;   // z, broadcast source for the post-check mat-vec
    __syncthreads();

    // ---- post-check against the ORIGINAL blocks, src/avi.jl:71-76 / :148-156 -----------
    double rk = act ? a.q[vo + lane] : 0.0;
    for (int j = 0; j < N; ++j) {
        double zj = prow[j];
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
        double t = p - d;
        if (t < lk) t = lk;
        if (t > uk) t = uk;
        nres = fabs(p - t);
        if (isnan(nres)) nres = QINF;
        // comp_indices, src/avi_solutions.jl:511-562
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
}

} // namespace

int qpn_avi_max_n() { return qpn_avi_big_max_n(); }   // N <= 64: one wavefront; larger: qpn_avi_big.hip

// Kernel choice: the register-tableau kernel (qpn_avi_reg.hip) is the production path; the
// LDS-tableau kernel of this file stays selectable (QPN_AVI_KERNEL=lds1) for A/B measurements.
hipError_t qpn_launch_avi_solve(const AviBatchArgs &a, hipStream_t stream)
{
    static const int use_lds1 = [] {
        const char *e = QPN_DEV_ENV("QPN_AVI_KERNEL");
        return (e && strcmp(e, "lds1") == 0) ? 1 : 0;
    }();
    if (use_lds1) return qpn_launch_avi_solve_lds1(a, stream);
    // Items of shape [free STD x n | GAVI x m] (a node's reduced KKT system) take the MFMA
    // Schur-complement kernel; it flags everything else (and any item whose H block fails its
    // no-pivoting test) with status = -1, and the register kernel solves exactly those in a second,
    // gated launch.  Pure box-MCP batches (kind == NULL) and already gated launches go straight to
    // the register kernel.  QPN_AVI_KERNEL=reg forces the general kernel for A/B runs.
    static const int force_reg = [] {
        const char *e = QPN_DEV_ENV("QPN_AVI_KERNEL");
        return (e && strcmp(e, "reg") == 0) ? 1 : 0;
    }();
    if (force_reg || a.kind == nullptr || a.only_if != nullptr || a.N < 2) return qpn_launch_avi_solve_reg(a, stream);
    hipError_t e = qpn_launch_avi_solve_schur(a, nullptr, nullptr, nullptr, nullptr, stream);
    if (e != hipSuccess) return e;
    AviBatchArgs g = a;
    g.only_if = a.status; g.only_if_value = -1; g.scan = 1;
    return qpn_launch_avi_solve_reg(g, stream);
}

hipError_t qpn_launch_avi_solve_lds1(const AviBatchArgs &a, hipStream_t stream)
{
    if (a.batch <= 0) return hipSuccess;
    const LdsLayout L = lds_layout(a.N);
    static QpnPerDeviceOnce attr_once;
    const int attr_dev = attr_once.device();
    if (!attr_once.done[attr_dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(avi_solve_lds1),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_once.done[attr_dev] = true;
    }
    hipLaunchKernelGGL(avi_solve_lds1, dim3((unsigned)a.batch), dim3(WAVE), L.bytes, stream, a);
    return hipGetLastError();
}
