/*
 * qpn_oracle.h -- CPU restatement of the QPNet node-AVI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * What it restates (reference = forrestlaine/QuadraticProgramNetworks.jl v0.4.0,
 * paths relative to /root/reference):
 *   qpo_check_avi_solution   src/avi.jl:148-156
 *   qpo_solve_avi            src/avi.jl:63-77   (the PATHSolver.solve_mcp call + post check)
 *   qpo_convert_gavi         src/avi.jl:113-128
 *   qpo_assemble_node        src/avi.jl:205-251 + 305-377 (single-node pool, dense, reduced form)
 *   qpo_verify_solution      src/qp_processing.jl:57-149 (+ :12-33 bounded-LSQ fallback)
 *   qpo_comp_indices         src/avi_solutions.jl:511-562, 587-612
 *   qpo_local_piece          src/avi_solutions.jl:400-441, 491-496, 384-388 (piece of a recipe, before simplify)
 *
 * PARITY STATUS: "parity unpinned" at the solve_mcp boundary.  The arithmetic
 * behind src/avi.jl:64 lives in PATHSolver.jl (compat "1.7", Project.toml:28;
 * no Manifest, closed-source libpath), which is absent here and the reference
 * holds no golden vector at that boundary.  The pivotal method below restates
 * PATH's published algorithm class (Dirkse & Ferris 1995: crash basis + Lemke-type
 * complementary pivoting on the linearised box-MCP; one linearisation is exact for
 * an affine problem).  It is pinned end-to-end by the reference's own
 * test/simple_bilevel.jl:4-21 cases (tests/test_oracle_golden.py) and by the
 * hand-derived AVI known answers of SURVEY.md section 8(c).
 *
 * All matrices are dense, COLUMN-MAJOR (Julia layout), fp64.  +-INFINITY bounds
 * are literal IEEE infinities.
 */
#ifndef QPN_ORACLE_H
#define QPN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: the reference enum, src/avi.jl:1-6 */
enum { QPO_SUCCESS = 1, QPO_RAY_TERM = 2, QPO_MAX_ITERS = 3, QPO_FAILURE = 4 };

/* row kinds of the mixed problem
 *   QPO_ROW_STD : (Mz+q)_i  complementary to  l_i <= z_i <= u_i      (AVI row, src/avi.jl:56-61)
 *   QPO_ROW_GAVI:  z_i      complementary to  l_i <= (Mz+q)_i <= u_i (second GAVI condition, src/avi.jl:22-24)
 */
enum { QPO_ROW_STD = 0, QPO_ROW_GAVI = 1 };

typedef struct {
    double check_tol;   /* 1e-6, src/avi.jl:148                         */
    double piv_tol;     /* smallest admissible pivot magnitude          */
    double feas_tol;    /* basic infeasibility treated as zero          */
    int    max_pivots;  /* <=0: 50*N+100                                */
} qpo_opts;

void qpo_default_opts(qpo_opts *o);

/* src/avi.jl:148-156 (generalised to GAVI rows by swapping the roles of z_i and r_i).
 * Returns the violation count ("degree"); r_out (may be NULL) receives Mz+q. */
int qpo_check_avi_solution(int N, const double *M, const double *q, const double *l,
                           const double *u, const uint8_t *rowkind, const double *z,
                           double tol, double *r_out);

/* natural-map residual  max_i | p_i - proj_[l,u](p_i - d_i) |  */
double qpo_natural_residual(int N, const double *M, const double *q, const double *l,
                            const double *u, const uint8_t *rowkind, const double *z);

/* One AVI: z holds z0 on entry, the solution on exit.  rowkind may be NULL (all STD).
 * active (may be NULL) receives the comp_indices masks at tol 1e-2.  Returns status. */
int qpo_solve_avi(int N, const double *M, const double *q, const double *l, const double *u,
                  const uint8_t *rowkind, double *z, const qpo_opts *opts,
                  double *resid_out, int *pivots_out, uint8_t *active);

/* Batch of independent AVIs.  strideM = 0 shares one M across the batch.
 * nthreads <= 0: all cores (OpenMP).  Returns number of items with status != SUCCESS. */
int qpo_solve_avi_batch(int batch, int N, const double *M, long strideM, const double *q,
                        const double *l, const double *u, const uint8_t *rowkind,
                        long stride_kind, double *z, const qpo_opts *opts, int32_t *status,
                        double *resid, int32_t *pivots, uint8_t *active, int nthreads);

/* src/avi.jl:113-128: GAVI (d1,d2) -> box AVI of size d1+2*d2.
 * M is d1 x (d1+d2), A is d2 x (d1+d2); outputs sized (d1+2d2)^2 and d1+2d2. */
void qpo_convert_gavi(int d1, int d2, const double *M, const double *o, const double *l1,
                      const double *u1, const double *A, const double *bw, const double *l2,
                      const double *u2, double *Mout, double *qout, double *lout, double *uout);

/* Single-node pool, reduced form:
 *   Mout = [[Qd, -Ad'],[Ad, 0]]  (N = n+m), qout = [qd + R w ; B w], bounds on the lambda rows,
 *   kind = [STD x n ; GAVI x m],  lout/uout = [-inf/+inf x n ; l ; u]. */
void qpo_assemble_node(int n, int m, int p, const double *Qd, const double *R, const double *qd,
                       const double *Ad, const double *B, const double *l, const double *u,
                       const double *w, double *Mout, double *qout, double *lout, double *uout,
                       uint8_t *kind);

/* src/avi_solutions.jl:511-562: per-row code mask, bit (c-1) set for code c in 1..4.
 * shift = 0 for the z1 block, 4 for the s2 block of the GAVI wrapper (:587-612). */
void qpo_comp_indices(int n, const double *zv, const double *rv, const double *l, const double *u,
                      double tol, int shift, uint8_t *mask);

/* src/qp_processing.jl:57-149 on a dense node record.
 * Returns 1 if the point is optimal for the node (solution=true), 0 otherwise.
 * lambda (length m) receives the multipliers (sign: + at lower bound, - at upper, :120-123).
 * path_out: 0 = infeasible, 1 = m==0 shortcut, 2 = least-squares duals accepted,
 *           3 = bounded-LSQ fallback accepted, 4 = fallback rejected, 5 = fallback solver failed. */
int qpo_verify_solution(int n, int m, int p, const double *Qd, const double *R, const double *qd,
                        const double *Ad, const double *B, const double *l, const double *u,
                        const double *xd, const double *w, double tol, double *lambda,
                        int *path_out);

/* src/avi_solutions.jl:400-441 + :491-496 (find_non_trivial :384-388), reducible_inds empty (the live caller, `expand`,
 * :246-247), for the per-node GAVI of process_solution_graph (src/avi.jl:447-477): z = [x_d (n); lambda (m)], w = x_p (p),
 *   M = [Qd -Ad'], N = R, o = qd, l1/u1 = -+inf, A = [Ad 0], B = B, l2 = l, u2 = u.
 * K[i] in 1..4 for i < n, 5..8 for i >= n: the recipe's code of row i (:390-399).
 * Outputs the piece BEFORE simplify (polyhedral; host): Ap [(2N) x (N+p)] column-major, N = n+m, rows
 * [M N ; I2 0 ; I1 0 ; A B], bounds lp/up [2N] (noisy l > u fixed to l = u, :437-438; entries <= 1e-8 dropped, :439),
 * keep [2N] = find_non_trivial (a finite bound and a non-empty row). */
void qpo_local_piece(int n, int m, int p, const double *Qd, const double *R, const double *qd, const double *Ad,
                     const double *B, const double *l, const double *u, const uint8_t *K, double *Ap, double *lp,
                     double *up, uint8_t *keep);

int qpo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
