/*
 * qpn_oracle.c -- CPU restatement of the QPNet node-AVI hot path (see qpn_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, loaded by, or called from the product
 * (quadraticprogramnetworks.jl_amd/).  "parity unpinned" at the PATHSolver.solve_mcp
 * boundary (src/avi.jl:64-70): see the header for what pins it instead.
 *
 * The pivotal AVI method (qpo_solve_avi), in the vocabulary used throughout:
 *   every index k carries a complementary PAIR (p_k, d_k):
 *       STD  row: p_k = z_k,       d_k = (Mz+q)_k      (src/avi.jl:56-61)
 *       GAVI row: p_k = (Mz+q)_k,  d_k = z_k           (src/avi.jl:22-24, second condition)
 *   with  l_k <= p_k <= u_k,  d_k >= 0 if p_k = l_k,  d_k <= 0 if p_k = u_k,  d_k = 0 inside.
 *   A dictionary (tableau)  basic = T * nonbasic  is kept, exactly one member of each pair
 *   basic.  Stage A ("crash", cf. PATH's crash + src/deprecated/avi_scratch.jl:30-51) brings
 *   every free variable into the basis by partial pivoting; Stage B is Lemke's
 *   complementary pivoting with a covering column built from the initial infeasibility
 *   (cf. the residual column r of src/deprecated/avi_scratch.jl:21-23) and bounded-variable
 *   ratio tests (:63-77), ending when the artificial leaves (SUCCESS), nothing blocks
 *   (RAY_TERM) or the pivot budget is spent (MAX_ITERS).  The post-check of
 *   src/avi.jl:71-76 turns any violated answer into FAILURE.
 */
#include "qpn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QINF INFINITY

void qpo_default_opts(qpo_opts *o)
{
    o->check_tol = 1e-6;
    o->piv_tol = 1e-11;
    o->feas_tol = 1e-12;
    o->max_pivots = 0;
}

int qpo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* src/avi.jl:148-156                                                   */
/* ------------------------------------------------------------------ */
static void matvec_q(int N, const double *M, const double *q, const double *z, double *r)
{
    for (int i = 0; i < N; ++i) r[i] = q[i];
    for (int j = 0; j < N; ++j) {
        double zj = z[j];
        if (zj == 0.0) continue;
        const double *col = M + (size_t)j * N;
        for (int i = 0; i < N; ++i) r[i] = fma(col[i], zj, r[i]);
    }
}

int qpo_check_avi_solution(int N, const double *M, const double *q, const double *l,
                           const double *u, const uint8_t *rowkind, const double *z,
                           double tol, double *r_out)
{
    double *r = r_out ? r_out : (double *)malloc(sizeof(double) * (size_t)N);
    matvec_q(N, M, q, z, r);
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        int g = rowkind ? rowkind[i] : 0;
        double p = g ? r[i] : z[i]; /* the bounded member   */
        double d = g ? z[i] : r[i]; /* its multiplier       */
        if (d > tol && fabs(p - l[i]) > tol) bad++;   /* :152 first term  */
        if (d < -tol && fabs(p - u[i]) > tol) bad++;  /* :153             */
        if (p - l[i] < -tol) bad++;                   /* :154             */
        if (p - u[i] > tol) bad++;
        if (isnan(p) || isnan(d)) bad++;
    }
    if (!r_out) free(r);
    return bad;
}

double qpo_natural_residual(int N, const double *M, const double *q, const double *l,
                            const double *u, const uint8_t *rowkind, const double *z)
{
    double *r = (double *)malloc(sizeof(double) * (size_t)N);
    matvec_q(N, M, q, z, r);
    double res = 0.0;
    for (int i = 0; i < N; ++i) {
        int g = rowkind ? rowkind[i] : 0;
        double p = g ? r[i] : z[i];
        double d = g ? z[i] : r[i];
        double t = p - d;
        if (t < l[i]) t = l[i];
        if (t > u[i]) t = u[i];
        double e = fabs(p - t);
        if (isnan(e)) e = QINF;
        if (e > res) res = e;
    }
    free(r);
    return res;
}

/* ------------------------------------------------------------------ */
/* src/avi_solutions.jl:511-562                                         */
/* ------------------------------------------------------------------ */
static int approx_eq(double a, double b, double tol)
{ /* Julia isapprox(a,b;atol=tol) for scalars (rtol = 0 when atol > 0) */
    if (a == b) return 1;
    if (!isfinite(a) || !isfinite(b)) return 0;
    return fabs(a - b) <= tol;
}

void qpo_comp_indices(int n, const double *zv, const double *rv, const double *l, const double *u,
                      double tol, int shift, uint8_t *mask)
{
    for (int i = 0; i < n; ++i) {
        int eq = approx_eq(l[i], u[i], tol);           /* :512 */
        int rz = fabs(rv[i]) <= tol;                   /* :513 */
        unsigned m = 0;
        if (!eq) {
            if (approx_eq(zv[i], l[i], tol) && rv[i] >= -tol) m |= 1u;               /* :543 */
            if (l[i] - tol <= zv[i] && zv[i] <= u[i] + tol && rz) m |= 2u;           /* :546 */
            if (approx_eq(zv[i], u[i], tol) && rv[i] <= tol) m |= 4u;                /* :549 */
        } else {
            m = 8u;                                                                  /* :552-558 */
        }
        mask[i] = (uint8_t)(m << shift);
    }
}

/* ------------------------------------------------------------------ */
/* the pivotal solver                                                   */
/* ------------------------------------------------------------------ */
typedef struct {
    int N, NC;
    double *T;       /* N x NC, row-major: basic_i = sum_j T[i][j] * nonbasic_j (+const, kept in xb) */
    double *xb;      /* current value of the basic variable of each row   */
    double *nbval;   /* current value of the nonbasic variable of each col */
    int *rowvar;     /* variable id basic in row i                        */
    int *colvar;     /* variable id nonbasic in column j                  */
    int *posb;       /* var id -> row, or -1                              */
    int *posn;       /* var id -> col, or -1                              */
    uint8_t *at_upper;
    const double *l, *u;
    double piv_tol;
} tab_t;

static inline int pair_free(const tab_t *S, int k) { return S->l[k] == -QINF && S->u[k] == QINF; }
static inline int pair_fixed(const tab_t *S, int k) { return S->l[k] == S->u[k]; }

/* admissible interval of variable v while it is basic */
static void var_interval(const tab_t *S, int v, double *lo, double *hi)
{
    int N = S->N;
    if (v == 2 * N) { *lo = 0.0; *hi = QINF; return; }
    if (v < N) { *lo = S->l[v]; *hi = S->u[v]; return; }
    int k = v - N;
    if (pair_fixed(S, k)) { *lo = -QINF; *hi = QINF; return; }
    if (pair_free(S, k)) { *lo = 0.0; *hi = 0.0; return; }
    if (S->at_upper[k]) { *lo = -QINF; *hi = 0.0; }
    else { *lo = 0.0; *hi = QINF; }
}

/* move the entering variable (column c) by delta, then exchange it with row r;
 * the leaving variable is parked at leave_val. */
static void do_pivot(tab_t *S, int r, int c, double delta, double leave_val)
{
    int N = S->N, NC = S->NC;
    double *T = S->T;
    for (int i = 0; i < N; ++i) S->xb[i] = fma(delta, T[(size_t)i * NC + c], S->xb[i]);
    double enter_val = S->nbval[c] + delta;

    double piv = T[(size_t)r * NC + c];
    double inv = 1.0 / piv;
    double *Tr = T + (size_t)r * NC;
    for (int j = 0; j < NC; ++j) Tr[j] = Tr[j] * inv; /* prow_j; entry c fixed below */
    for (int i = 0; i < N; ++i) {
        if (i == r) continue;
        double *Ti = T + (size_t)i * NC;
        double cm = Ti[c];
        double ncm = -cm;
        for (int j = 0; j < NC; ++j) Ti[j] = fma(ncm, Tr[j], Ti[j]); /* entry c rewritten below */
        Ti[c] = cm * inv;
    }
    for (int j = 0; j < NC; ++j)
        if (j != c) Tr[j] = -Tr[j];
    Tr[c] = inv;

    int ve = S->colvar[c], vl = S->rowvar[r];
    S->rowvar[r] = ve; S->posb[ve] = r; S->posn[ve] = -1;
    S->colvar[c] = vl; S->posn[vl] = c; S->posb[vl] = -1;
    S->xb[r] = enter_val;
    S->nbval[c] = leave_val;
}

static int is_must_leave(const tab_t *S, int v, double *target)
{
    int N = S->N;
    if (v >= 2 * N) return 0;
    if (v >= N) { /* multiplier of a free variable: an equation, must sit at 0 */
        if (pair_free(S, v - N)) { *target = 0.0; return 1; }
        return 0;
    }
    if (pair_fixed(S, v) && !pair_free(S, v)) { *target = S->l[v]; return 1; }
    return 0;
}

static int is_ordinary_pair(const tab_t *S, int k) { return !pair_free(S, k) && !pair_fixed(S, k); }

/* bytes of scratch one solve of size N needs */
static size_t avi_ws_bytes(int N)
{
    size_t NC = (size_t)N + 1;
    size_t d = (size_t)N * NC + N + NC + N;              /* T, xb, nbval, r */
    size_t i = (size_t)N + NC + 2 * (2 * (size_t)N + 1) + (8 * (size_t)N + 8);
    return d * sizeof(double) + i * sizeof(int) + (size_t)N + 64;
}

static int solve_avi_ws(int N, const double *M, const double *q, const double *l, const double *u,
                        const uint8_t *rowkind, double *z, const qpo_opts *opts_in,
                        double *resid_out, int *pivots_out, uint8_t *active, void *ws)
{
    qpo_opts opts;
    if (opts_in) opts = *opts_in; else qpo_default_opts(&opts);
    int max_piv = opts.max_pivots > 0 ? opts.max_pivots : 50 * N + 100;
    int NC = N + 1;
    int status = QPO_FAILURE;
    int pivots = 0;

    tab_t S;
    S.N = N; S.NC = NC; S.l = l; S.u = u; S.piv_tol = opts.piv_tol;
    double *dws = (double *)ws;
    S.T = dws; dws += (size_t)N * NC;
    S.xb = dws; dws += N;
    S.nbval = dws; dws += NC;
    double *r = dws; dws += N;
    int *iws = (int *)dws;
    S.rowvar = iws; iws += N;
    S.colvar = iws; iws += NC;
    S.posb = iws; iws += 2 * N + 1;
    S.posn = iws; iws += 2 * N + 1;
    int *enter_list = iws; iws += 8 * N + 8;
    S.at_upper = (uint8_t *)iws;
    memset(S.at_upper, 0, (size_t)N);
    int n_enter = 0;

    for (int v = 0; v <= 2 * N; ++v) { S.posb[v] = -1; S.posn[v] = -1; }

    /* ---- initial dictionary: basic = (Mz+q)_i, nonbasic = z_j ---- */
    for (int j = 0; j < N; ++j) {
        int g = rowkind ? rowkind[j] : 0;
        double v0;
        if (g) { /* GAVI row: z_j is the multiplier, nonbasic at 0 (duals cold, src/avi.jl:404) */
            v0 = 0.0;
            S.colvar[j] = N + j; S.rowvar[j] = j;
        } else {
            double z0 = z[j];
            if (isnan(z0)) z0 = 0.0;
            if (pair_free(&S, j)) v0 = z0;
            else {
                double lo = l[j], hi = u[j];
                if (z0 < lo) z0 = lo;
                if (z0 > hi) z0 = hi;
                if (lo == -QINF) { v0 = hi; S.at_upper[j] = 1; }
                else if (hi == QINF) { v0 = lo; }
                else if (hi - z0 < z0 - lo) { v0 = hi; S.at_upper[j] = 1; }
                else v0 = lo;
                if (hi == lo) S.at_upper[j] = 0;
            }
            S.colvar[j] = j; S.rowvar[j] = N + j;
        }
        S.nbval[j] = v0;
        S.posn[S.colvar[j]] = j;
        S.posb[S.rowvar[j]] = j;
    }
    S.colvar[N] = 2 * N; S.posn[2 * N] = N; S.nbval[N] = 0.0;
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) S.T[(size_t)i * NC + j] = M[(size_t)j * N + i];
        S.T[(size_t)i * NC + N] = 0.0;
    }
    matvec_q(N, M, q, S.nbval, S.xb);

    /* scale of the problem: max |M_ij|; a diagonal crash pivot must reach 1e-4 of it (error growth <= 1e4 eps) */
    double mscale = 0.0;
    for (size_t t = 0; t < (size_t)N * N; ++t) { double a = fabs(M[t]); if (a > mscale) mscale = a; }
    const double diag_thr = 1e-4 * (mscale > 1.0 ? mscale : 1.0);

    /* ---- Stage A: crash.  Free variables (and multipliers of equality GAVI rows) enter. ---- */
    for (int k = 0; k < N; ++k) {
        int g = rowkind ? rowkind[k] : 0;
        if (!g && pair_free(&S, k)) enter_list[n_enter++] = k;
        if (g && pair_fixed(&S, k)) enter_list[n_enter++] = N + k;
    }
    int stageA_budget = 4 * N + 4;
    for (int idx = 0; idx < n_enter && stageA_budget > 0; ++idx) {
        int e = enter_list[idx];
        int c = S.posn[e];
        if (c < 0) continue;
        double colmax = 0.0, best = 0.0; int r = -1; double target = 0.0;
        for (int i = 0; i < N; ++i) {
            double a = fabs(S.T[(size_t)i * NC + c]);
            if (a > colmax) colmax = a;
            double tg;
            if (is_must_leave(&S, S.rowvar[i], &tg) && a > best) { best = a; r = i; target = tg; }
        }
        double thresh = 1e-9 * (colmax > 1.0 ? colmax : 1.0);
        /* diagonal first: a free variable whose column is still its own takes its own equation
         * row when that pivot is not tiny against the scale of M -- no search (elimination
         * without pivoting is stable for the (semi)definite blocks QP nodes produce);
         * otherwise the largest admissible row (partial pivoting). */
        if (e < N && c == e && S.rowvar[e] == N + e) {
            double ad = fabs(S.T[(size_t)e * NC + c]);
            if (ad >= diag_thr) { r = e; best = ad; target = 0.0; thresh = 0.0; }
        }
        if (r >= 0 && best > thresh) {
            double delta = (target - S.xb[r]) * (1.0 / S.T[(size_t)r * NC + c]);
            do_pivot(&S, r, c, delta, target);
            pivots++; stageA_budget--;
            continue;
        }
        /* no equation row can take it: exchange with an ordinary pair and queue that pair's
         * other member (a 2x2 principal block pivot, e.g. LP-like nodes with Q = 0). */
        best = 0.0; r = -1;
        for (int i = 0; i < N; ++i) {
            int v = S.rowvar[i];
            if (v >= 2 * N) continue;
            int k = v < N ? v : v - N;
            if (!is_ordinary_pair(&S, k)) continue;
            double a = fabs(S.T[(size_t)i * NC + c]);
            if (a > best) { best = a; r = i; }
        }
        if (r < 0 || best <= thresh) continue; /* stays nonbasic; Stage B copes or the check fails */
        int v = S.rowvar[r];
        if (v < N) {
            double x = S.xb[r], lo = l[v], hi = u[v];
            if (x <= lo) { target = lo; S.at_upper[v] = 0; }
            else if (x >= hi) { target = hi; S.at_upper[v] = 1; }
            else if (lo == -QINF) { target = hi; S.at_upper[v] = 1; }
            else if (hi == QINF) { target = lo; S.at_upper[v] = 0; }
            else if (hi - x < x - lo) { target = hi; S.at_upper[v] = 1; }
            else { target = lo; S.at_upper[v] = 0; }
            enter_list[n_enter++] = N + v;
        } else {
            target = 0.0;
            enter_list[n_enter++] = v - N;
        }
        double delta = (target - S.xb[r]) * (1.0 / S.T[(size_t)r * NC + c]);
        do_pivot(&S, r, c, delta, target);
        pivots++; stageA_budget--;
        if (n_enter >= 8 * N) break;
    }

    /* ---- Stage B: covering column from the basic infeasibilities ---- */
    double theta0 = 0.0;
    for (int i = 0; i < N; ++i) {
        double lo, hi; var_interval(&S, S.rowvar[i], &lo, &hi);
        double x = S.xb[i];
        double v = 0.0;
        if (x < lo) v = lo - x; else if (x > hi) v = x - hi;
        if (v > theta0) theta0 = v;
    }
    if (theta0 <= opts.feas_tol) {
        status = QPO_SUCCESS;
    } else {
        for (int i = 0; i < N; ++i) {
            double lo, hi; var_interval(&S, S.rowvar[i], &lo, &hi);
            double x = S.xb[i], cov = 0.0;
            if (x < lo) {
                double tgt = lo + (theta0 - (lo - x));
                if (hi < QINF) { double mid = 0.5 * (lo + hi); if (tgt > mid) tgt = mid; }
                cov = (tgt - x) / theta0; S.xb[i] = tgt;
            } else if (x > hi) {
                double tgt = hi - (theta0 - (x - hi));
                if (lo > -QINF) { double mid = 0.5 * (lo + hi); if (tgt < mid) tgt = mid; }
                cov = (tgt - x) / theta0; S.xb[i] = tgt;
            }
            S.T[(size_t)i * NC + N] = cov;
        }
        S.nbval[N] = theta0;

        int c = N;            /* entering column: the artificial, decreasing from theta0 */
        double sigma = -1.0;  /* direction of travel of the entering variable            */
        double self_lim = theta0;
        status = QPO_MAX_ITERS;
        while (pivots < max_piv) {
            /* ratio test, two passes (Harris): pass 1 finds the step bound with slack,
             * pass 2 takes the largest pivot among the rows inside it. */
            const double slack = 1e-10;
            double dmax = self_lim;
            for (int i = 0; i < N; ++i) {
                double g = sigma * S.T[(size_t)i * NC + c];
                double lo, hi; var_interval(&S, S.rowvar[i], &lo, &hi);
                double d;
                if (g < -S.piv_tol && lo > -QINF) { double rc = 1.0 / g; d = (S.xb[i] - lo) * (-rc) + slack * (-rc); }
                else if (g > S.piv_tol && hi < QINF) { double rc = 1.0 / g; d = (hi - S.xb[i]) * rc + slack * rc; }
                else continue;
                if (d < dmax) dmax = d;
            }
            if (dmax == QINF) { status = QPO_RAY_TERM; break; }
            int r = -1; double bestg = 0.0, step = 0.0, leave_val = 0.0;
            int theta_row = S.posb[2 * N];
            for (int i = 0; i < N; ++i) {
                double g = sigma * S.T[(size_t)i * NC + c];
                double lo, hi; var_interval(&S, S.rowvar[i], &lo, &hi);
                double d, lv;
                if (g < -S.piv_tol && lo > -QINF) { d = (S.xb[i] - lo) * (-(1.0 / g)); lv = lo; }
                else if (g > S.piv_tol && hi < QINF) { d = (hi - S.xb[i]) * (1.0 / g); lv = hi; }
                else continue;
                if (d > dmax) continue;
                double ag = fabs(g);
                if (i == theta_row) ag = QINF; /* let the artificial leave whenever it can */
                if (ag > bestg) { bestg = ag; r = i; step = d < 0.0 ? 0.0 : d; leave_val = lv; }
            }
            if (r < 0) {
                /* the entering variable reaches its own far bound first */
                double delta = sigma * self_lim;
                for (int i = 0; i < N; ++i)
                    S.xb[i] = fma(delta, S.T[(size_t)i * NC + c], S.xb[i]);
                int ve = S.colvar[c];
                if (ve == 2 * N) { S.nbval[c] = 0.0; status = QPO_SUCCESS; break; }
                /* ve is a bounded member p_k flipping bound; its multiplier enters next */
                int k = ve;
                S.at_upper[k] = sigma > 0 ? 1 : 0;
                S.nbval[c] = S.at_upper[k] ? u[k] : l[k];
                pivots++;
                int vn = N + k;
                c = S.posn[vn];
                if (c < 0) { status = QPO_FAILURE; break; }
                sigma = S.at_upper[k] ? -1.0 : 1.0;
                self_lim = QINF;
                continue;
            }
            int vl = S.rowvar[r];
            do_pivot(&S, r, c, sigma * step, leave_val);
            pivots++;
            if (vl == 2 * N) { status = QPO_SUCCESS; break; }
            int vn;
            if (vl < N) { /* bounded member left at a bound: its multiplier enters */
                int k = vl;
                if (!pair_fixed(&S, k)) S.at_upper[k] = (leave_val == u[k]) ? 1 : 0;
                vn = N + k;
                sigma = S.at_upper[k] ? -1.0 : 1.0;
                self_lim = QINF;
            } else {      /* multiplier hit 0: the bounded member leaves its bound */
                int k = vl - N;
                vn = k;
                sigma = S.at_upper[k] ? -1.0 : 1.0;
                self_lim = u[k] - l[k];
                if (pair_free(&S, k)) { self_lim = QINF; sigma = 1.0; }
            }
            c = S.posn[vn];
            if (c < 0) { status = QPO_FAILURE; break; }
        }
    }

    /* ---- read the point back ---- */
    for (int k = 0; k < N; ++k) {
        int g = rowkind ? rowkind[k] : 0;
        int vz = g ? N + k : k; /* which member of the pair is z_k */
        double val = S.posb[vz] >= 0 ? S.xb[S.posb[vz]] : S.nbval[S.posn[vz]];
        z[k] = val;
    }

    /* ---- post-check, src/avi.jl:71-76 ---- */
    int bad = qpo_check_avi_solution(N, M, q, l, u, rowkind, z, opts.check_tol, r);
    if (bad > 0) { if (status == QPO_SUCCESS) status = QPO_FAILURE; }
    if (resid_out) { /* natural-map residual from the r = Mz+q just computed */
        double res = 0.0;
        for (int i = 0; i < N; ++i) {
            int g = rowkind ? rowkind[i] : 0;
            double p = g ? r[i] : z[i], d = g ? z[i] : r[i];
            double t = p - d;
            if (t < l[i]) t = l[i];
            if (t > u[i]) t = u[i];
            double e = fabs(p - t);
            if (isnan(e)) e = QINF;
            if (e > res) res = e;
        }
        *resid_out = res;
    }
    if (pivots_out) *pivots_out = pivots;
    if (active) {
        for (int k = 0; k < N; ++k) {
            int g = rowkind ? rowkind[k] : 0;
            double p = g ? r[k] : z[k], d = g ? z[k] : r[k];
            qpo_comp_indices(1, &p, &d, l + k, u + k, 1e-2, g ? 4 : 0, active + k);
        }
    }
    return status;
}

int qpo_solve_avi(int N, const double *M, const double *q, const double *l, const double *u,
                  const uint8_t *rowkind, double *z, const qpo_opts *opts_in,
                  double *resid_out, int *pivots_out, uint8_t *active)
{
    void *ws = malloc(avi_ws_bytes(N));
    int st = solve_avi_ws(N, M, q, l, u, rowkind, z, opts_in, resid_out, pivots_out, active, ws);
    free(ws);
    return st;
}

int qpo_solve_avi_batch(int batch, int N, const double *M, long strideM, const double *q,
                        const double *l, const double *u, const uint8_t *rowkind,
                        long stride_kind, double *z, const qpo_opts *opts, int32_t *status,
                        double *resid, int32_t *pivots, uint8_t *active, int nthreads)
{
    int nfail = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads) reduction(+ : nfail)
#endif
    {
        void *ws = malloc(avi_ws_bytes(N)); /* one scratch block per thread, reused */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int b = 0; b < batch; ++b) {
            double res; int piv;
            int st = solve_avi_ws(N, M + (size_t)b * strideM, q + (size_t)b * N, l + (size_t)b * N,
                                  u + (size_t)b * N,
                                  rowkind ? rowkind + (size_t)b * stride_kind : NULL,
                                  z + (size_t)b * N, opts, &res, &piv,
                                  active ? active + (size_t)b * N : NULL, ws);
            if (status) status[b] = st;
            if (resid) resid[b] = res;
            if (pivots) pivots[b] = piv;
            if (st != QPO_SUCCESS) nfail++;
        }
        free(ws);
    }
    (void)nthreads;
    return nfail;
}

/* ------------------------------------------------------------------ */
/* src/avi.jl:113-128                                                   */
/* ------------------------------------------------------------------ */
void qpo_convert_gavi(int d1, int d2, const double *M, const double *o, const double *l1,
                      const double *u1, const double *A, const double *bw, const double *l2,
                      const double *u2, double *Mout, double *qout, double *lout, double *uout)
{
    int dz = d1 + d2, N = d1 + 2 * d2;
    memset(Mout, 0, sizeof(double) * (size_t)N * N);
    for (int j = 0; j < dz; ++j) {
        for (int i = 0; i < d1; ++i) Mout[(size_t)j * N + i] = M[(size_t)j * d1 + i];          /* [M 0]    */
        for (int i = 0; i < d2; ++i) Mout[(size_t)j * N + d1 + i] = A[(size_t)j * d2 + i];     /* [A -I]   */
    }
    for (int i = 0; i < d2; ++i) {
        Mout[(size_t)(dz + i) * N + d1 + i] = -1.0;
        Mout[(size_t)(d1 + i) * N + dz + i] = 1.0;                                           /* [0 I 0]  */
    }
    for (int i = 0; i < d1; ++i) { qout[i] = o[i]; lout[i] = l1[i]; uout[i] = u1[i]; }
    for (int i = 0; i < d2; ++i) {
        qout[d1 + i] = bw ? bw[i] : 0.0; lout[d1 + i] = -QINF; uout[d1 + i] = QINF;
        qout[dz + i] = 0.0; lout[dz + i] = l2[i]; uout[dz + i] = u2[i];
    }
}

/* ------------------------------------------------------------------ */
/* src/avi.jl:205-251 + 305-377, single-node pool, reduced form          */
/* ------------------------------------------------------------------ */
void qpo_assemble_node(int n, int m, int p, const double *Qd, const double *R, const double *qd,
                       const double *Ad, const double *B, const double *l, const double *u,
                       const double *w, double *Mout, double *qout, double *lout, double *uout,
                       uint8_t *kind)
{
    int N = n + m;
    memset(Mout, 0, sizeof(double) * (size_t)N * N);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) Mout[(size_t)j * N + i] = Qd[(size_t)j * n + i];
        for (int i = 0; i < m; ++i) {
            double a = Ad[(size_t)j * m + i];
            Mout[(size_t)j * N + n + i] = a;          /* A block            */
            Mout[(size_t)(n + i) * N + j] = -a;       /* -A' block          */
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = qd[i];
        for (int k = 0; k < p; ++k) s = fma(R[(size_t)k * n + i], w[k], s);
        qout[i] = s; lout[i] = -QINF; uout[i] = QINF; kind[i] = QPO_ROW_STD;
    }
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int k = 0; k < p; ++k) s = fma(B[(size_t)k * m + i], w[k], s);
        qout[n + i] = s; lout[n + i] = l[i]; uout[n + i] = u[i]; kind[n + i] = QPO_ROW_GAVI;
    }
}

/* ------------------------------------------------------------------ */
/* src/qp_processing.jl:57-149                                          */
/* ------------------------------------------------------------------ */

/* least squares  min || C y - b ||  for C (n x k, column-major) by Householder QR with
 * column pivoting; rank-deficient columns get y = 0 (a basic solution, as the sparse
 * QR behind Julia's `\` at :115 returns). */
static void lsq_qrcp(int n, int k, double *C, double *b, double *y)
{
    int *perm = (int *)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
    double *cn = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    double *dsc = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    /* the columns are brought to unit length first (y is scaled back at the end): the pivot order and the rank decision
     * then do not depend on how a constraint row happens to be scaled -- one row of length 1e7 among rows of length 1
     * must not push the others under the rank threshold.  (The device kernels do the same on their Gram block.) */
    for (int j = 0; j < k; ++j) {
        perm[j] = j;
        double s = 0; for (int i = 0; i < n; ++i) s += C[(size_t)j * n + i] * C[(size_t)j * n + i];
        dsc[j] = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
        for (int i = 0; i < n; ++i) C[(size_t)j * n + i] *= dsc[j];
        cn[j] = s > 0.0 ? 1.0 : 0.0;
    }
    int rank = 0; int steps = n < k ? n : k;
    double scale = 0; for (int j = 0; j < k; ++j) if (cn[j] > scale) scale = cn[j];
    for (int s = 0; s < steps; ++s) {
        int jm = s; double best = -1;
        for (int j = s; j < k; ++j) {
            double t = 0; for (int i = s; i < n; ++i) t += C[(size_t)j * n + i] * C[(size_t)j * n + i];
            cn[j] = t; if (t > best) { best = t; jm = j; }
        }
        if (best <= 1e-24 * (scale > 1 ? scale : 1)) break;
        /* norms within 2^-30 of the largest count as equal and the lowest column wins: after the equilibration every column
         * starts at length 1 up to rounding, and that rounding must not choose the pivot (it decides which multipliers of a
         * rank-deficient block are 0).  The device kernels apply the same rule (PIV_BAND in qpn_verify.hip). */
        jm = -1;                                     /* (lowest ORIGINAL column: perm[], positions move with the swaps) */
        for (int j = s; j < k; ++j) if (cn[j] >= best * (1.0 - 0x1p-30) && (jm < 0 || perm[j] < perm[jm])) jm = j;
        best = cn[jm];
        if (jm != s) {
            for (int i = 0; i < n; ++i) { double t = C[(size_t)s * n + i]; C[(size_t)s * n + i] = C[(size_t)jm * n + i]; C[(size_t)jm * n + i] = t; }
            int t = perm[s]; perm[s] = perm[jm]; perm[jm] = t;
        }
        double *v = C + (size_t)s * n;
        double alpha = sqrt(best); if (v[s] > 0) alpha = -alpha;
        double v0 = v[s] - alpha;
        double vnorm2 = best - v[s] * v[s] + v0 * v0;
        if (vnorm2 > 0) {
            /* apply H = I - 2 vv'/v'v to the remaining columns and b; v = (v0, v[s+1..]) */
            for (int j = s + 1; j < k; ++j) {
                double *cj = C + (size_t)j * n;
                double d = v0 * cj[s]; for (int i = s + 1; i < n; ++i) d += v[i] * cj[i];
                d = 2 * d / vnorm2;
                cj[s] -= d * v0; for (int i = s + 1; i < n; ++i) cj[i] -= d * v[i];
            }
            double d = v0 * b[s]; for (int i = s + 1; i < n; ++i) d += v[i] * b[i];
            d = 2 * d / vnorm2;
            b[s] -= d * v0; for (int i = s + 1; i < n; ++i) b[i] -= d * v[i];
        }
        v[s] = alpha; /* R diagonal; the sub-diagonal part is no longer needed */
        rank++;
    }
    double *yy = (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double));
    for (int s = rank - 1; s >= 0; --s) {
        double t = b[s];
        for (int j = s + 1; j < rank; ++j) t -= C[(size_t)j * n + s] * yy[j];
        yy[s] = t / C[(size_t)s * n + s];
    }
    for (int j = 0; j < k; ++j) y[perm[j]] = j < rank ? yy[j] * dsc[perm[j]] : 0.0;
    free(yy); free(perm); free(cn); free(dsc);
}

int qpo_verify_solution(int n, int m, int p, const double *Qd, const double *R, const double *qd,
                        const double *Ad, const double *B, const double *l, const double *u,
                        const double *xd, const double *w, double tol, double *lambda,
                        int *path_out)
{
    int path = 0, sol = 0;
    double *qt = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *ax = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    /* :58-60  q~ = Q[dec,:] x + q[dec] */
    for (int i = 0; i < n; ++i) {
        double s = qd[i];
        for (int j = 0; j < n; ++j) s = fma(Qd[(size_t)j * n + i], xd[j], s);
        for (int k = 0; k < p; ++k) s = fma(R[(size_t)k * n + i], w[k], s);
        qt[i] = s;
    }
    /* :84 */
    for (int i = 0; i < m; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) s = fma(Ad[(size_t)j * m + i], xd[j], s);
        for (int k = 0; k < p; ++k) s = fma(B[(size_t)k * m + i], w[k], s);
        ax[i] = s;
    }
    for (int i = 0; i < m; ++i) lambda[i] = 0.0;
    /* :86  feasibility with tol 1e-3 (Slice membership, src/sets.jl:851-854) */
    int feasible = 1;
    for (int i = 0; i < m; ++i)
        if (!(l[i] - 1e-3 <= ax[i] && ax[i] - 1e-3 <= u[i])) feasible = 0;
    if (!feasible) { path = 0; goto done; }
    if (m == 0) { /* :91-96 */
        double s = 0; for (int i = 0; i < n; ++i) s += qt[i] * qt[i];
        path = 1; sol = sqrt(s) <= tol;
        goto done;
    }
    {
        /* :98-103 */
        uint8_t *cls = (uint8_t *)malloc(m > 0 ? (size_t)m : 1); /* 0 none, 1 pos, 2 neg, 3 both */
        int np = 0, nn = 0, nb = 0;
        for (int i = 0; i < m; ++i) {
            int pos = ax[i] < l[i] + 1e-2, neg = ax[i] > u[i] - 1e-2;
            cls[i] = (uint8_t)((pos ? 1 : 0) | (neg ? 2 : 0));
            if (cls[i] == 1) np++; else if (cls[i] == 2) nn++; else if (cls[i] == 3) nb++;
        }
        int k = np + nn + nb;
        /* :114  A_bar = [A+' -A-' A0']  (n x k) */
        double *C = (double *)malloc(sizeof(double) * (size_t)n * (k > 0 ? k : 1));
        int *rowof = (int *)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
        int c = 0;
        for (int pass = 1; pass <= 3; ++pass)
            for (int i = 0; i < m; ++i)
                if (cls[i] == pass) {
                    double sg = pass == 2 ? -1.0 : 1.0;
                    for (int j = 0; j < n; ++j) C[(size_t)c * n + j] = sg * Ad[(size_t)j * m + i];
                    rowof[c++] = i;
                }
        double *Cw = (double *)malloc(sizeof(double) * (size_t)n * (k > 0 ? k : 1));
        double *bw = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        double *y = (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double));
        memcpy(Cw, C, sizeof(double) * (size_t)n * k);
        memcpy(bw, qt, sizeof(double) * (size_t)n);
        lsq_qrcp(n, k, Cw, bw, y); /* :115 */
        int ok = 1;
        for (int j = 0; j < np + nn; ++j) if (!(y[j] > -tol)) ok = 0;            /* :119 signs   */
        double rs = 0;
        for (int i = 0; i < n; ++i) {
            double s = -qt[i];
            for (int j = 0; j < k; ++j) s += C[(size_t)j * n + i] * y[j];
            rs += s * s;
        }
        if (!(sqrt(rs) <= tol)) ok = 0;                                           /* :119 residual */
        if (ok) {
            for (int j = 0; j < k; ++j) {
                int i = rowof[j];
                lambda[i] = cls[i] == 2 ? -y[j] : y[j];                           /* :120-123 */
            }
            path = 2; sol = 1;
        } else {
            /* :129-137  min 1/2 ||Ad' lam - q~||^2, sign bounds; PATH MCP of :12-27 reduced to
             * the box-AVI  lb <= lam <= ub  complementary to  (Ad Ad') lam - Ad q~           */
            double *G = (double *)malloc(sizeof(double) * (size_t)m * m);
            double *gq = (double *)malloc(sizeof(double) * (size_t)m);
            double *lb = (double *)malloc(sizeof(double) * (size_t)m);
            double *ub = (double *)malloc(sizeof(double) * (size_t)m);
            for (int i = 0; i < m; ++i) {
                for (int j = 0; j < m; ++j) {
                    double s = 0;
                    for (int t = 0; t < n; ++t) s = fma(Ad[(size_t)t * m + i], Ad[(size_t)t * m + j], s);
                    G[(size_t)j * m + i] = s;
                }
                double s = 0;
                for (int t = 0; t < n; ++t) s = fma(Ad[(size_t)t * m + i], qt[t], s);
                gq[i] = -s;
                lb[i] = (cls[i] & 2) ? -QINF : 0.0;  /* :129-131 */
                ub[i] = (cls[i] & 1) ? QINF : 0.0;   /* :132-134 */
                lambda[i] = 0.0;
            }
            qpo_opts o; qpo_default_opts(&o);
            double res; int piv;
            int st = qpo_solve_avi(m, G, gq, lb, ub, NULL, lambda, &o, &res, &piv, NULL);
            if (st != QPO_SUCCESS) { path = 5; sol = 0; }
            else {
                double rs2 = 0;
                for (int t = 0; t < n; ++t) {
                    double s = -qt[t];
                    for (int i = 0; i < m; ++i) s += Ad[(size_t)t * m + i] * lambda[i];
                    rs2 += s * s;
                }
                if (sqrt(rs2) <= 1e-4) { path = 3; sol = 1; }                      /* :138 */
                else { path = 4; sol = 0; }
            }
            free(G); free(gq); free(lb); free(ub);
        }
        free(cls); free(C); free(rowof); free(Cw); free(bw); free(y);
    }
done:
    if (path_out) *path_out = path;
    free(qt); free(ax);
    return sol;
}

/* ---- local_piece, src/avi_solutions.jl:400-441 + :491-496 --------------------------------------------------------
 * A = [gavi.M gavi.N; I2 0; I1 0; gavi.A gavi.B] (:405-408) for the per-node GAVI of src/avi.jl:466-473; bounds per
 * recipe code (:413-432); l = [bounds[:,1]; bounds[:,3]], u = [bounds[:,2]; bounds[:,4]] (:434-435). */
void qpo_local_piece(int n, int m, int p, const double *Qd, const double *R, const double *qd, const double *Ad,
                     const double *B, const double *l, const double *u, const uint8_t *K, double *Ap, double *lp,
                     double *up, uint8_t *keep)
{
    const int N = n + m, rows = 2 * N, cols = N + p;
    const double inf = INFINITY;
    for (long t = 0; t < (long)rows * cols; ++t) Ap[t] = 0.0;
#define AP(r, c) Ap[(size_t)(c) * rows + (r)]
    for (int i = 0; i < n; ++i) {                       /* [M N]: row i = [Qd(i,:)  -Ad(:,i)'  R(i,:)] */
        for (int j = 0; j < n; ++j) AP(i, j) = Qd[(size_t)j * n + i];
        for (int k = 0; k < m; ++k) AP(i, n + k) = -Ad[(size_t)i * m + k];
        for (int c = 0; c < p; ++c) AP(i, N + c) = R[(size_t)c * n + i];
    }
    for (int k = 0; k < m; ++k) AP(n + k, n + k) = 1.0;                 /* [I2 0] */
    for (int i = 0; i < n; ++i) AP(N + i, i) = 1.0;                     /* [I1 0] */
    for (int k = 0; k < m; ++k) {                                       /* [A B], A = [Ad 0] */
        for (int j = 0; j < n; ++j) AP(N + n + k, j) = Ad[(size_t)j * m + k];
        for (int c = 0; c < p; ++c) AP(N + n + k, N + c) = B[(size_t)c * m + k];
    }
    for (int i = 0; i < N; ++i) {
        double b1, b2, b3, b4;
        const int code = K[i];
        if (i < n) {
            const double o = qd[i], l1 = -inf, u1 = inf;
            switch (code) {
            case 1: b1 = -o; b2 = inf; b3 = l1; b4 = l1; break;
            case 2: b1 = -o; b2 = -o; b3 = l1; b4 = u1; break;
            case 3: b1 = -inf; b2 = -o; b3 = u1; b4 = u1; break;
            default: b1 = -inf; b2 = inf; b3 = l1; b4 = u1; break;       /* 4 */
            }
        } else {
            const double l2 = l[i - n], u2 = u[i - n];
            switch (code) {
            case 5: b1 = 0.0; b2 = inf; b3 = l2; b4 = l2; break;
            case 6: b1 = 0.0; b2 = 0.0; b3 = l2; b4 = u2; break;
            case 7: b1 = -inf; b2 = 0.0; b3 = u2; b4 = u2; break;
            default: b1 = -inf; b2 = inf; b3 = l2; b4 = u2; break;       /* 8 */
            }
        }
        lp[i] = b1; up[i] = b2; lp[N + i] = b3; up[N + i] = b4;
    }
    for (int r = 0; r < rows; ++r) {
        if (lp[r] > up[r]) lp[r] = up[r];                                /* noisy_inds, :437-438 */
        int nz = 0;
        for (int c = 0; c < cols; ++c) {
            if (fabs(AP(r, c)) <= 1e-8) AP(r, c) = 0.0;                  /* droptol!, :439 */
            else nz = 1;
        }
        keep[r] = (uint8_t)((!isinf(lp[r]) || !isinf(up[r])) && nz);     /* find_non_trivial, :384-388 */
    }
#undef AP
}
