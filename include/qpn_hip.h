/*
 * qpn_hip.h -- C-ABI of the MI355X-native QPNet node-AVI engine (libqpn_hip.so).
 *
 * Drop-in boundary for the hot path of forrestlaine/QuadraticProgramNetworks.jl v0.4.0
 * (paths below are relative to the reference tree).  Every entry point is `extern "C"`,
 * takes plain pointers and sizes, returns an int error code (0 = ok, <0 = API misuse or HIP
 * error; per-item solver outcomes are ONLY reported in status[]), never throws, never
 * keeps a caller pointer past return, and keeps no global state (one qpn_ctx per
 * thread/stream is safe).  All matrices are dense COLUMN-MAJOR fp64 (Julia's layout);
 * +-Inf bounds are literal IEEE infinities (src/avi.jl:125-126).
 *
 * `mem` says where the caller's buffers live: QPN_MEM_HOST (the library stages them
 * through HBM itself -- what the Julia `ccall` shim uses) or QPN_MEM_DEVICE (pointers are
 * already in HBM on the ctx's device; the call is asynchronous on the ctx stream).
 */
#ifndef QPN_HIP_H
#define QPN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPN_ABI_VERSION 1

/* per-item status: the reference enum StatusCode, src/avi.jl:1-6 */
enum { QPN_SUCCESS = 1, QPN_RAY_TERM = 2, QPN_MAX_ITERS = 3, QPN_FAILURE = 4 };

/* API error codes (function return values) */
enum {
    QPN_OK = 0,
    QPN_ERR_ARG = -1,      /* bad argument (null pointer, size out of range)  */
    QPN_ERR_HIP = -2,      /* a HIP runtime call failed; see qpn_ctx_last_error */
    QPN_ERR_NODEVICE = -3, /* no gfx950 device visible                        */
    QPN_ERR_SIZE = -4      /* problem size not supported by any kernel        */
};

enum { QPN_MEM_HOST = 0, QPN_MEM_DEVICE = 1 };

/* row kinds of the mixed complementarity problem solved per item
 *   QPN_ROW_STD : (Mz+q)_i  _|_  l_i <= z_i <= u_i       AVI row,  src/avi.jl:56-61
 *   QPN_ROW_GAVI:  z_i      _|_  l_i <= (Mz+q)_i <= u_i  second GAVI condition, src/avi.jl:22-24
 * A batch with kind == NULL is a plain box-MCP: exactly PATHSolver.solve_mcp's problem. */
enum { QPN_ROW_STD = 0, QPN_ROW_GAVI = 1 };

typedef struct qpn_ctx qpn_ctx;

/* qpn_avi_opts.flags */
#define QPN_AVI_FLAG_COLD_START 1 /* ignore the z input: every item starts from z0 = 0 (no reset pass) */

typedef struct {
    double check_tol;  /* post-check tolerance, 1e-6       (src/avi.jl:148)                 */
    double piv_tol;    /* smallest admissible pivot, 1e-11                                   */
    double feas_tol;   /* basic infeasibility treated as zero, 1e-12                         */
    double comp_tol;   /* active-set classification tolerance, 1e-2 (src/avi_solutions.jl:511) */
    int32_t max_pivots; /* <= 0: 50*N + 100 (cf. PATH limits at src/avi.jl:67-70)             */
    int32_t flags;      /* QPN_AVI_FLAG_* (0 by default)                                        */
} qpn_avi_opts;

/* ---- context ------------------------------------------------------------------- */
int qpn_abi_version(void);
int qpn_ctx_create(int device_id, qpn_ctx **out);
int qpn_ctx_destroy(qpn_ctx *ctx);
/* Launch on an existing hipStream_t (e.g. torch's current stream).  NULL is HIP's legacy default
 * (null) stream, NOT "no stream".  A new ctx launches on a private non-blocking stream;
 * qpn_ctx_use_own_stream returns to it. */
int qpn_ctx_set_stream(qpn_ctx *ctx, void *hip_stream);
int qpn_ctx_use_own_stream(qpn_ctx *ctx);
int qpn_ctx_synchronize(qpn_ctx *ctx);
const char *qpn_ctx_last_error(qpn_ctx *ctx);
const char *qpn_strerror(int code);
void qpn_avi_default_opts(qpn_avi_opts *opts);

/* ---- (A2+A3+A9) batched AVI solve ------------------------------------------------
 * Replaces PATHSolver.solve_mcp as called at src/avi.jl:64-70 and src/qp_processing.jl:22-27,
 * including the post-check of src/avi.jl:71-76 (check_avi_solution, :148-156) and the
 * active-set classification of src/avi_solutions.jl:511-562 / :587-612.
 *   M       [batch][N*N] column-major, item stride strideM doubles (0: one M shared by all items)
 *   q,l,u   [batch][N]
 *   kind    [batch][N] uint8 row kinds, item stride stride_kind (0: shared), or NULL (all STD)
 *   z       [batch][N] in: z0 (warm start of the bounded STD variables; duals start cold as
 *                       src/avi.jl:404)   out: solution
 *   status  [batch] int32 (QPN_SUCCESS..QPN_FAILURE), resid [batch] natural-map residual,
 *   pivots  [batch] int32, active [batch][N] uint8: bit (c-1) set for code c of
 *           src/avi_solutions.jl:511-562 on STD rows, bit (c+3) on GAVI rows (codes 5..8).
 *   Any of resid/pivots/active may be NULL.  N <= 1024 in ABI version 1 (N <= 64: one wavefront per item; larger: one workgroup). */
int qpn_solve_avi_batch(qpn_ctx *ctx, int32_t batch, int32_t N, const double *M, int64_t strideM,
                        const double *q, const double *l, const double *u, const uint8_t *kind,
                        int64_t stride_kind, double *z, int32_t *status, double *resid,
                        int32_t *pivots, uint8_t *active, const qpn_avi_opts *opts, int mem);

/* One problem in Julia's own SparseMatrixCSC{Float64,Int32} layout (1-based colptr/rowval):
 * the exact argument list of PATHSolver.solve_mcp(M, q, l, u, z0) at src/avi.jl:64.
 * Host pointers only.  z: in z0, out solution. */
int qpn_solve_mcp_csc(qpn_ctx *ctx, int32_t N, const int32_t *colptr, const int32_t *rowval,
                      const double *nzval, const double *q, const double *l, const double *u,
                      double *z, int32_t *status, double *resid, int32_t *pivots,
                      const qpn_avi_opts *opts);

/* ---- (A3) batched check_avi_solution, src/avi.jl:148-156 --------------------------
 *   degree [batch] int32 violation count (sol_bad = degree > 0), r [batch][N] = Mz+q (may be NULL) */
int qpn_check_avi_batch(qpn_ctx *ctx, int32_t batch, int32_t N, const double *M, int64_t strideM,
                        const double *q, const double *l, const double *u, const uint8_t *kind,
                        int64_t stride_kind, const double *z, double tol, int32_t *degree,
                        double *r, int mem);

/* ---- (A9) comp_indices core, src/avi_solutions.jl:511-562 --------------------------
 * Flat arrays of `count` rows; mask bit (c-1+shift) for code c in 1..4 (shift = 4 for the
 * s2 block of the GAVI wrapper, :587-612). */
int qpn_comp_indices(qpn_ctx *ctx, int64_t count, const double *zv, const double *rv,
                     const double *l, const double *u, double tol, int32_t shift, uint8_t *mask,
                     int mem);

/* ---- (A5+A6) per-node KKT assembly, single-node pools, reduced form -----------------
 * src/avi.jl:205-251 + :305-377 restated densely without the dead xi block and the slack
 * block: per node i with n decision variables, m constraint rows,
 * p parameters,
 *     M_i = [[Qd_i, -Ad_i'],[Ad_i, 0]]   (N = n+m),   q_i = [qd_i + R_i w ; B_i w],
 *     l/u = [-Inf/+Inf x n ; l_i ; u_i],  kind = [STD x n ; GAVI x m].
 *   Qd [batch][n*n], R [batch][n*p], qd [batch][n], Ad [batch][m*n], B [batch][m*p],
 *   l,u [batch][m], w [batch][p] with item stride stride_w (0: one shared parameter vector). */
int qpn_assemble_nodes(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p,
                       const double *Qd, const double *R, const double *qd, const double *Ad,
                       const double *B, const double *l, const double *u, const double *w,
                       int64_t stride_w, double *Mout, double *qout, double *lout, double *uout,
                       uint8_t *kind_out, int mem);

/* ---- (A6) pool assembly: combine_gavis, src/avi.jl:305-377 (and convert, :113-128) -------------------------------
 * One AVI per Nash pool (all nodes of a level jointly, src/avi.jl:399-400), `batch` instances of ONE pool shape
 * (config 3: 1 000 payoff draws of the four-player game; a level of a net: batch = 1).  The shape (host arrays, read
 * during the call only): `players` nodes in pool order (sorted ids, :319), player i with n_i decision variables and m_i
 * constraint rows; the pool's decision variables are the union (nd positions = dec_inds, sorted), the other variables
 * of the net are the p parameters; dpos (concatenated over the players, sum n_i entries) = position in dec_inds of each
 * of a player's decision variables.
 * Numeric inputs, the players' blocks stacked in pool order, column-major, sn = sum n_i, sm = sum m_i:
 *   Qd [sn x nd] = Q_i[dvars_i, dec_inds]     Qp [sn x p] = Q_i[dvars_i, param_inds]     qd [sn] = q_i[dvars_i]
 *   Ad [sm x nd] = M2_i[:, dec_inds]          Bp [sm x p] = M2_i[:, param_inds]          l, u [sm]
 *   w [p] parameters.  Every input has an item stride in doubles; 0 shares it across the batch.
 * Output: M [N x N] column-major (item stride strideM; 0 = ONE shared M, allowed when Qd and Ad are shared), q, l, u,
 * kind [batch][N], ready for qpn_solve_avi_batch.
 *   QPN_POOL_REFERENCE  z = [dec | xi_i per player | lambda-psi_i per player | slack], N = nd + sn + 2 sm: the AVI the
 *                       reference hands to PATH, rows [sum_i xi^i_d = 0 (:356-367) | player KKT rows (:335-340) |
 *                       [A -I] | [0 I 0] (:113-128)], all STD
 *   QPN_POOL_REDUCED    z = [dec | lambda-psi], N = nd + sm, kinds [STD x nd | GAVI x sm]: without the xi block (it is
 *                       multiplied by 0, :244) and the slack block; needs disjoint decision sets (sn = nd)
 * qpn_pool_size returns N for a shape and form. */
typedef struct {
    int32_t players, nd, p;
    const int32_t *n_i, *m_i; /* [players] */
    const int32_t *dpos;      /* [sum n_i] */
} qpn_pool_shape;
enum { QPN_POOL_REDUCED = 0, QPN_POOL_REFERENCE = 1 };
int qpn_pool_size(const qpn_pool_shape *shape, int form, int32_t *N);
int qpn_assemble_pools(qpn_ctx *ctx, const qpn_pool_shape *shape, int form, int32_t batch, const double *Qd,
                       int64_t stride_Qd, const double *Qp, int64_t stride_Qp, const double *qd, int64_t stride_qd,
                       const double *Ad, int64_t stride_Ad, const double *Bp, int64_t stride_Bp, const double *l,
                       const double *u, int64_t stride_lu, const double *w, int64_t stride_w, double *Mout,
                       int64_t strideM, double *qout, double *lout, double *uout, uint8_t *kind_out, int mem);

/* ---- (A5+A6+A2+A3+A9) fused: assemble each node's KKT blocks on the fly and solve ----------
 * Same inputs as qpn_assemble_nodes, same outputs as qpn_solve_avi_batch (N = n+m, z = [x_d; lambda]);
 * identical results to calling the two in sequence, without materialising M in HBM: one pass of the
 * hot path per outer sweep (src/algorithm.jl:95 -> solve_qep -> src/avi.jl:399-409 for single-node
 * pools).  z: in z0 (ignored with QPN_AVI_FLAG_COLD_START), out solution.  n, m <= 32 run on the fused
 * matrix-core kernel (one wavefront per node); n, m <= 64 on the four-wavefronts-per-node path (crash on the
 * matrix cores straight from the records, cold duals); larger nodes (n+m <= 1024) are assembled and take the
 * blocked matrix-core path for large node-shaped items (n, m <= 512) or the general kernels. */
int qpn_solve_nodes(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                    const double *R, const double *qd, const double *Ad, const double *B,
                    const double *l, const double *u, const double *w, int64_t stride_w, double *z,
                    int32_t *status, double *resid, int32_t *pivots, uint8_t *active,
                    const qpn_avi_opts *opts, int mem);

/* Same, and additionally scatters every node's primal block x_d = z[0..n) into the caller's iterate:
 *   x[b * stride_x + i] = z[b][i], i < n   (stride_x >= n, in doubles; x in the same memory space as z).
 * This is the write-back the outer sweep does after each solve (src/algorithm.jl:97-101,
 * x[decision_inds] = x_opt[decision_inds]); doing it from the solve kernel spares two copy kernels per
 * sweep.  x == NULL behaves exactly like qpn_solve_nodes. */
int qpn_solve_nodes_into(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                         const double *R, const double *qd, const double *Ad, const double *B,
                         const double *l, const double *u, const double *w, int64_t stride_w, double *z,
                         int32_t *status, double *resid, int32_t *pivots, uint8_t *active,
                         const qpn_avi_opts *opts, int mem, double *x, int64_t stride_x);

/* ---- resident node records: upload once, sweep many times ----------------------------------------------
 * The outer loop (src/algorithm.jl:13-117) sweeps the SAME nodes again and again: between two sweeps only the
 * parameters w (the other players' decision variables) change, while Qd, R, qd, Ad, B, l, u -- 22 KB per
 * n = m = 32 node -- stay what they were.  A caller with host arrays (the Julia shim) that went through
 * qpn_solve_nodes every sweep would move those records over PCIe every time (2.3 M solves/s against 85 M/s
 * from resident records).  qpn_nodes_upload copies the records into HBM owned by the
 * library ONCE and returns a handle; qpn_solve_nodes_h / qpn_verify_nodes_h then take the handle, w and the
 * output buffers, exactly as qpn_solve_nodes_into / qpn_verify_nodes would with the records in place.
 *   mem (upload)    where Qd..u live (QPN_MEM_HOST / QPN_MEM_DEVICE); the handle holds its own copy either way,
 *                   so the records cannot change under it
 *   mem (solve)     where w, z0/z, status, resid, pivots, active, x live.  With QPN_MEM_HOST only w goes up and
 *                   only the requested outputs come down; z may be NULL when only the primal blocks (x) or only
 *                   the statuses are wanted.
 * The handle also keeps what depends on the records alone: whether any of its nodes needs the general
 * (pivoting) kernel -- decided by Qd, Ad, l, u, never by w -- so that sweeps over well-conditioned nodes are ONE
 * launch, and the longest-first schedule of its nodes (from exponentially smoothed pivot counts, which every sweep's
 * solve kernel updates; the order is re-sorted from them every `period` sweeps at first and every 4 x `period` once
 * settled; qpn_nodes_set_schedule, period 0 = natural order).
 * qpn_nodes_update replaces one array of the records (e.g. the bounds after a new child piece was chosen). */
typedef struct qpn_nodes qpn_nodes;
enum { QPN_NODE_QD = 0, QPN_NODE_R = 1, QPN_NODE_Q = 2, QPN_NODE_AD = 3, QPN_NODE_B = 4, QPN_NODE_L = 5, QPN_NODE_U = 6 };
int qpn_nodes_upload(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p, const double *Qd,
                     const double *R, const double *qd, const double *Ad, const double *B, const double *l,
                     const double *u, int mem, qpn_nodes **out);
int qpn_nodes_update(qpn_ctx *ctx, qpn_nodes *nodes, int32_t field, const double *data, int mem);
int qpn_nodes_set_schedule(qpn_ctx *ctx, qpn_nodes *nodes, int32_t period);
int qpn_nodes_free(qpn_ctx *ctx, qpn_nodes *nodes);
/* info[0] = what is known about the general kernel's share of these records: 0 nothing yet, 1 the count of the first
 * sweep is on its way to the host, 2 no node needs it (sweeps are one launch), 3 some do; info[1] = that count (valid
 * in states 2, 3); info[2] = bit 0: a longest-first schedule is installed, bit 1: every Qd block of the records is bitwise
 * symmetric (settled by one pass when the records arrive or Qd is replaced; QPN_OPT_SYM_ROUTE); info[3] = sweeps since the
 * last schedule reset. */
int qpn_nodes_info(qpn_ctx *ctx, qpn_nodes *nodes, int32_t info[4]);
int qpn_solve_nodes_h(qpn_ctx *ctx, qpn_nodes *nodes, const double *w, int64_t stride_w, double *z,
                      int32_t *status, double *resid, int32_t *pivots, uint8_t *active,
                      const qpn_avi_opts *opts, int mem, double *x, int64_t stride_x);
int qpn_verify_nodes_h(qpn_ctx *ctx, qpn_nodes *nodes, const double *xd, const double *w, int64_t stride_w,
                       double tol, int32_t *solution, double *lambda, int32_t *path, int mem);

/* ---- schedule hint for qpn_solve_nodes[_into]: longest solves first ---------------------------------
 * The outer loop (src/algorithm.jl:13-117) sweeps the SAME nodes again and again, and a node's pivot
 * count changes little between sweeps, while a launch ends with a tail in which the last, longest
 * solves run on a nearly empty GPU.  qpn_order_nodes_by_pivots builds, on the device, a permutation of
 * 0..count-1 by DESCENDING pivot count (`pivots` = the output of an earlier sweep over the same nodes)
 * and installs it in the context; later qpn_solve_nodes[_into] calls with batch == count hand node
 * order[i] to the i-th wavefront.  Inputs, outputs and their layout do not change -- only which
 * wavefront solves which node (results are independent of it, bit for bit).  qpn_set_node_order installs
 * a caller-made permutation (entries outside 0..count-1 leave their slot unsolved) or, with order ==
 * NULL, clears the hint.  The hint is ignored whenever batch != count. */
int qpn_order_nodes_by_pivots(qpn_ctx *ctx, const int32_t *pivots, int32_t count, int mem);
int qpn_set_node_order(qpn_ctx *ctx, const int32_t *order, int32_t count, int mem);
/* By default the context does this by itself: a qpn_solve_nodes[_into] call that fills the GPU (batch > 4096, n, m <= 32)
 * and returns pivot counts refreshes the hint from them every `period`-th call of the same batch size (default 16; the
 * first call of a batch size installs it), one 8 us launch behind the solve.  A hint installed through the two
 * functions above takes precedence until qpn_set_node_order(ctx, NULL, ...) clears it; period = 0 switches the
 * mechanism off.  Results never depend on any of this. */
int qpn_ctx_set_auto_schedule(qpn_ctx *ctx, int32_t period);

/* Per-context options that select between kernel routes with IDENTICAL contracts (A/B measurements, tests of a route
 * against the one it replaced).  Results never depend on them beyond rounding.
 *   QPN_OPT_MID_ROUTE  node records with n, m <= 128 and one of them > 32 (qpn_solve_nodes*): 1 = one fused kernel per node
 *                      (default: ONE wavefront per node for max(n, m) <= 48, one workgroup per node beyond), 0 = the general
 *                      route (assembled blocks -> the route of large nodes / the general kernels): the cross-check of the tests.
 *                      (Round 3's values 2 and 3 -- round 2's three-kernel route, the workgroup kernel for 33 .. 48 -- are gone;
 *                      the library reads no environment variable; diagnostic builds with -DQPN_DEV_SWITCHES accept a preset.)
 *   QPN_OPT_BIG_ROUTE  kept for callers that set the default: takes 1 only (the blocked matrix-core crash straight from the
 *                      records, BASELINE config 5); round 2's route over an assembled M is gone as a node-record switch -- the
 *                      same kernels serve large node-shaped items passed as M through qpn_solve_avi_batch.
 *   QPN_OPT_SYM_ROUTE  resident records (qpn_nodes_upload) whose Qd blocks are ALL bitwise symmetric: 1 = kernel variants that
 *                      use the symmetry of H and of S = A H^-1 A' (default; n = m = 32: 8 of 88 fp64 MFMAs per solve less; large
 *                      nodes, 64 < n <= 256 and m <= 256: the complementarity phase runs as block principal pivoting on the
 *                      symmetric Schur problem, the Lemke kernel behind it for what it leaves -- `pivots` then reports n + the
 *                      number of complementarity pairs switched; with a caller-set max_pivots the Lemke kernel, whose pivots
 *                      that budget counts, runs alone), 0 = the general variants.  Records with any asymmetric Qd, and records
 *                      passed per call, always take the general variants.
 * A resident handle remembers under which option values it learned that none of its nodes declines; after a change it asks again
 * on its next sweep (another kernel variant applies its pivot test to slightly different numbers). */
#define QPN_OPT_MID_ROUTE 1
#define QPN_OPT_BIG_ROUTE 2
#define QPN_OPT_SYM_ROUTE 3
int qpn_ctx_set_option(qpn_ctx *ctx, int32_t option, int32_t value);

/* ---- multi-GPU: replicas of the iterate on peer GPUs, written by the solve itself -----------------------
 * One process per GPU; rank g solves its own node range and every rank needs the whole iterate x for the
 * next sweep (src/algorithm.jl:95-101 reads x_opt of all children).  Instead of a collective after the solve,
 * the solve kernel stores each primal block to the local iterate AND to the same offset of the peers'
 * iterates: xGMI is point-to-point, the blocks are 8n bytes, and the stores ride along with the launch.
 *   qpn_shared_alloc   zeroed device buffer on ctx's GPU + its 64-byte IPC handle, to be sent to the peers by any
 *                      host channel (torch.distributed all_gather_object, MPI, a pipe, ...)
 *   qpn_shared_open    map a peer's buffer into this process (hipIpcOpenMemHandle); qpn_shared_close unmaps
 *   qpn_shared_free    release a buffer from qpn_shared_alloc (peers must have closed it)
 *   qpn_set_primal_mirrors  own = this rank's iterate buffer ([bytes], from qpn_shared_alloc), peers[k] = the
 *                      k-th peer's buffer as opened here (count <= QPN_MAX_MIRRORS; count = 0 clears).  Later
 *                      qpn_solve_nodes_into calls (device memory) whose x lies inside own[] also write
 *                      peers[k] + (x - own).  The stores are complete when the launch is; the CALLER orders
 *                      them against the peers' next reads (one barrier / tiny all-reduce per sweep). */
#define QPN_MAX_MIRRORS 7
#define QPN_IPC_HANDLE_BYTES 64
#define QPN_SHARED_FINE_GRAINED 1 /* qpn_shared_alloc flag: fine-grained (in-kernel cross-GPU visibility), for mailboxes */
int qpn_shared_alloc(qpn_ctx *ctx, size_t bytes, int flags, void **dev_ptr, uint8_t *handle);
int qpn_shared_open(qpn_ctx *ctx, const uint8_t *handle, void **dev_ptr);
int qpn_shared_close(qpn_ctx *ctx, void *dev_ptr);
int qpn_shared_free(qpn_ctx *ctx, void *dev_ptr);
int qpn_set_primal_mirrors(qpn_ctx *ctx, const double *own, size_t bytes, int32_t count, double *const *peers);

/* Per-sweep stop / raise decision (the reference ends a sweep with solved = false when any solve of the level
 * failed, src/algorithm.jl:95-109, src/avi.jl:426): out[0] = number of items with status != QPN_SUCCESS,
 * out[1] = max resid (NaN if any), out[2] = 1; out has 4 doubles.  status/resid/out in device memory, one small
 * launch, no host sync.
 * With world > 1 the pair is combined over all ranks (sum, max) WITHOUT a collective: boxes[r] is rank r's
 * mailbox (QPN_SWEEP_BOX_BYTES, from qpn_shared_alloc(QPN_SHARED_FINE_GRAINED), opened here; boxes[rank] the own
 * one); each rank posts its pair into every mailbox and waits until its own holds all `world` posts of this
 * `epoch` (the caller counts sweeps: 1, 2, 3, ... -- identical on all ranks).  Because it is enqueued after the
 * solve on the same stream, it is also the barrier that orders the solve's replica stores
 * (qpn_set_primal_mirrors) against the peers' next reads.  A peer that does not arrive within timeout_ms gives
 * out[2] = 0 (out[0..1] then cover the ranks that did) and out[3] += 1 -- a sticky count of missed barriers that the
 * caller zeroes and reads whenever it likes. */
#define QPN_SWEEP_BOX_BYTES 512
int qpn_sweep_status(qpn_ctx *ctx, const int32_t *status, const double *resid, int32_t count, double *out,
                     int32_t rank, int32_t world, void *const *boxes, uint64_t epoch, int32_t timeout_ms);

/* ---- (F1) local pieces of a node's solution map: local_piece, src/avi_solutions.jl:400-496 -----------------------
 * For the per-node GAVI of process_solution_graph (src/avi.jl:447-477) -- z = [x_d (n); lambda (m)], w = x_p (p), built from
 * the same node records as qpn_solve_nodes -- and a recipe K (one code per row of z: 1..4 on the x_d rows, 5..8 on the
 * constraint rows, src/avi_solutions.jl:390-399; code 0 = no condition), the polyhedral piece on which that recipe holds:
 *     rows [M N ; I2 0 ; I1 0 ; A B] over [z; w]  (:405-408),  bounds per code (:413-432),  noisy l > u -> l = u (:437-438),
 *     entries <= 1e-8 dropped (:439),  keep[] = find_non_trivial (:384-388).
 * Output per piece: Ap [(2N) x (N+p)] column-major (N = n+m), lp, up [2N], keep [2N]; simplify / projection stay with the
 * caller (polyhedral algebra).  `pieces` items; item t uses the records of node node_of[t] (node_of == NULL: node t), so
 * the many recipes of one solution share its records (nodes = number of record sets behind the pointers).  node_of entries
 * outside 0 .. nodes-1: QPN_ERR_ARG for host arrays; for device arrays (not read by the host) the piece comes back EMPTY --
 * keep = 0 on every row, bounds (-inf, +inf), zero coefficients -- and no record is read.  Codes outside a row's range
 * (x_d rows: 1..3, constraint rows: 5..8) mean "no condition" like code 0 (what qpn_recipes_from_masks emits for an empty mask).
 * qpn_recipes_from_masks enumerates recipes from the active-set masks of a solve (`active` of qpn_solve_nodes, one uint8
 * per row: the code sets J of src/avi_solutions.jl:511-562): recipe number first + t of the Cartesian product of the rows'
 * code sets (all_Ks, :200-215; row 0 is the fastest digit) for t < count; *total (may be NULL) = number of recipes
 * (saturating at INT64_MAX). */
int qpn_local_pieces(qpn_ctx *ctx, int32_t pieces, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd,
                     const double *R, const double *qd, const double *Ad, const double *B, const double *l,
                     const double *u, const int32_t *node_of, const uint8_t *K, double *Ap, double *lp, double *up,
                     uint8_t *keep, int mem);
int qpn_recipes_from_masks(qpn_ctx *ctx, int32_t N, const uint8_t *mask, int64_t first, int32_t count, uint8_t *K,
                           int64_t *total, int mem);

/* ---- (F1, batched over a level) the solution-graph pieces of MANY nodes with O(1) calls -------------------------------------
 * The outer loop maps process_qp over the nodes of a level (src/algorithm.jl:44-52) and every optimal node below level 1
 * makes its solution graph (src/qp_processing.jl:158, :193-198, :231 -> process_solution_graph, src/avi.jl:447-477).
 * qpn_recipes_batch: all_Ks (src/avi_solutions.jl:200-215) for `nodes` solutions in one launch.  masks [nodes][N]; offsets
 *   [nodes + 1] (ALWAYS a host array, like a pool shape: offsets[0] = 0, node b gets the recipes 0 .. offsets[b+1]-offsets[b]-1 of
 *   its Cartesian product, at most as many as the product has); outputs K [offsets[nodes]][N] and node_of [offsets[nodes]] --
 *   exactly what qpn_local_pieces / qpn_reduced_pieces take.
 * qpn_reduced_pieces: local_piece (src/avi_solutions.jl:400-496, as qpn_local_pieces) followed by the elimination of the m
 *   multiplier columns through the piece's own equality rows (eliminate_variables, src/sets.jl:731-800: one column at a time,
 *   the alive equality row with the largest entry |a| > tol, the first of equal ones; that row then leaves) -- what
 *   project_and_permute (src/avi_solutions.jl:79-91) comes to when the active rows pin the multipliers.  Output per piece, over
 *   the columns [x_d (n); x_p (p)] and with room for cap = n + 2m rows: Ar = the cap x (n+p) matrix, column-major,
 *   lr, ur [cap], rows = the number of rows that remain (in their original order), flags: bit 0 = a multiplier column was pinned
 *   by no equality row while an alive row still holds it (degenerate active set: the projection needs Fourier-Motzkin / vertex
 *   enumeration, which stays with the caller -- the piece's output is then incomplete), bit 1 = more than cap rows remained.
 *   The local pieces themselves live in the context's workspace only.  n + m <= 512. */
int qpn_recipes_batch(qpn_ctx *ctx, int32_t nodes, int32_t N, const uint8_t *masks, const int64_t *offsets, uint8_t *K,
                      int32_t *node_of, int mem);
int qpn_reduced_pieces(qpn_ctx *ctx, int32_t pieces, int32_t nodes, int32_t n, int32_t m, int32_t p, const double *Qd,
                       const double *R, const double *qd, const double *Ad, const double *B, const double *l,
                       const double *u, const int32_t *node_of, const uint8_t *K, double tol, double *Ar, double *lr, double *ur,
                       int32_t *rows, int32_t *flags, int mem);

/* ---- (A8) batched per-node KKT verification, src/qp_processing.jl:57-149 ------------
 *   xd [batch][n] current decision values, w as above.
 *   solution [batch] int32 (1 = optimal for the node), lambda [batch][m] (sign: + at the lower
 *   bound, - at the upper, :120-123), path [batch] int32: 0 infeasible (:86-89), 1 m==0
 *   shortcut (:91-96), 2 least-squares duals accepted (:114-124), 3 bounded-LSQ fallback
 *   accepted (:129-139), 4 fallback rejected (:141), 5 fallback solver failed (:144).
 *   tol = 1e-4 (:57).  n, m <= 64: one wavefront per node; up to 512: one workgroup per node. */
int qpn_verify_nodes(qpn_ctx *ctx, int32_t batch, int32_t n, int32_t m, int32_t p,
                     const double *Qd, const double *R, const double *qd, const double *Ad,
                     const double *B, const double *l, const double *u, const double *xd,
                     const double *w, int64_t stride_w, double tol, int32_t *solution,
                     double *lambda, int32_t *path, int mem);

#ifdef __cplusplus
}
#endif
#endif /* QPN_HIP_H */
