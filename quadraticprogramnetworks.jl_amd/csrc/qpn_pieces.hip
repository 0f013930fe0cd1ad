// qpn_pieces.hip -- a LEVEL's worth of solution-graph pieces per call (batched over the nodes of
// a level: the per-node map of src/algorithm.jl:44-52 meets process_solution_graph, src/avi.jl:447-477).
//
//   recipes_batch_kernel   all_Ks (src/avi_solutions.jl:200-215) for MANY solutions: recipe t belongs to the node b with
//                          offsets[b] <= t < offsets[b+1] and is number t - offsets[b] of that node's Cartesian product
//   reduce_pieces_kernel   a local piece (local_piece, :400-496, made by local_pieces_kernel into a workspace) brought down from
//                          [x_d; lambda; x_p] to [x_d; x_p]: every multiplier column is eliminated through the piece's own
//                          equality rows -- the substitution of eliminate_variables (src/sets.jl:731-800), one column at a time,
//                          the alive equality row with the largest entry first -- which is what project_and_permute
//                          (src/avi_solutions.jl:79-91) comes to when the active rows pin the multipliers.  A column that no
//                          equality pins while an alive row still holds it (a degenerate active set: Fourier-Motzkin, polyhedral)
//                          raises the piece's flag and is left to the host.
//
// Arithmetic contract (the host restatement and the tests' CPU twin do the same operations in the same order, fp contraction
// off): f = A[k][j] / A[i][j];  A[k][c] = A[k][c] - f * A[i][c] (c != j), A[k][j] = 0;  l[k] = l[k] - f * l[i], u likewise.
#include "qpn_internal.h"

namespace {

constexpr double QINF = __builtin_huge_val();

__global__ __launch_bounds__(256) void recipes_batch_kernel(int32_t nodes, int32_t N, const uint8_t *masks, const long long *offsets,
                                                            long long total, uint8_t *K, int32_t *node_of)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int lo = 0, hi = nodes;                                          // the node b with offsets[b] <= t < offsets[b + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= t) lo = mid; else hi = mid;
    }
    const int b = lo;
    unsigned long long idx = (unsigned long long)(t - offsets[b]);
    const uint8_t *mask = masks + (size_t)b * N;
    for (int i = 0; i < N; ++i) {
        const unsigned mk = mask[i];
        const int radix = __popc(mk);
        int code = 0;
        if (radix > 0) {
            int d = (int)(idx % (unsigned)radix);
            idx /= (unsigned)radix;
            unsigned mm = mk;
            while (d-- > 0) mm &= mm - 1;                            // drop the d lowest set bits
            code = __ffs(mm);                                        // 1-based bit position = code
        }
        K[(size_t)t * N + i] = (uint8_t)code;
    }
    node_of[t] = b;
}

// One workgroup per piece.  A [rows x cols] column-major in the workspace (rows = 2N, cols = N + p), lp / up / keep [rows] as
// local_pieces_kernel left them.  Dynamic LDS: l, u, the current column (rows doubles each), the pivot row (cols doubles),
// alive flags and the output map (rows bytes / ints).
__global__ __launch_bounds__(256) void reduce_pieces_kernel(int32_t n, int32_t m, int32_t p, double tol, double *Ap, const double *lp,
                                                            const double *up, const uint8_t *keep, double *Ar, double *lr, double *ur,
                                                            int32_t *rows_out, int32_t *flags_out)
{
    extern __shared__ double lds[];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int N = n + m, rows = 2 * N, cols = N + p, cap = n + 2 * m;
    double *s_l = lds, *s_u = lds + rows, *s_col = lds + 2 * rows, *s_row = lds + 3 * rows;
    int *s_map = reinterpret_cast<int *>(s_row + cols);
    uint8_t *s_alive = reinterpret_cast<uint8_t *>(s_map + rows);
    __shared__ double r_val[256];
    __shared__ int r_row[256];
    __shared__ int s_flag, s_any, s_piv;
    double *A = Ap + (size_t)t * (size_t)rows * cols;
    for (int r = tid; r < rows; r += 256) {
        s_l[r] = lp[(size_t)t * rows + r]; s_u[r] = up[(size_t)t * rows + r]; s_alive[r] = keep[(size_t)t * rows + r];
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    for (int j = n; j < N; ++j) {
        // the alive equality row with the largest |A[., j]| (the first of equal ones)
        double best = 0.0; int brow = rows; int any = 0;
        for (int r = tid; r < rows; r += 256) {
            const double v = A[(size_t)j * rows + r];
            s_col[r] = v;
            const double a = fabs(v);
            if (s_alive[r]) {
                if (a > tol) any = 1;
                const double lo = s_l[r];
                if (lo == s_u[r] && !isinf(lo) && a > best) { best = a; brow = r; }   // (rows ascend per thread: the first maximum stays)
            }
        }
        r_val[tid] = best; r_row[tid] = brow;
        if (tid == 0) s_any = 0;
        __syncthreads();
        if (any) s_any = 1;                                          // (benign race: every writer stores 1)
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const double v2 = r_val[tid + s]; const int w2 = r_row[tid + s];
                if (v2 > r_val[tid] || (v2 == r_val[tid] && w2 < r_row[tid])) { r_val[tid] = v2; r_row[tid] = w2; }
            }
            __syncthreads();
        }
        const double bestv = r_val[0];
        const int i = r_row[0];
        if (!(bestv > tol)) {                                        // no equality pins lambda_j
            if (tid == 0 && s_any) s_flag = 1;
            __syncthreads();
            continue;
        }
        for (int c = tid; c < cols; c += 256) s_row[c] = A[(size_t)c * rows + i];
        __syncthreads();
        const double piv = s_col[i], li = s_l[i], ui = s_u[i];
        for (int k = tid; k < rows; k += 256) {
            if (k == i || !s_alive[k]) continue;
            const double akj = s_col[k];
            if (akj == 0.0) continue;
            const double f = akj / piv;
            for (int c = 0; c < cols; ++c) {
                const double prod = f * s_row[c];
                A[(size_t)c * rows + k] = A[(size_t)c * rows + k] - prod;
            }
            A[(size_t)j * rows + k] = 0.0;
            const double pl = f * li, pu = f * ui;
            s_l[k] = s_l[k] - pl; s_u[k] = s_u[k] - pu;
        }
        __syncthreads();
        if (tid == 0) s_alive[i] = 0;
        __syncthreads();
    }
    // the alive rows, in order, over the columns [x_d (n); x_p (p)]
    if (tid == 0) {
        int cnt = 0;
        for (int r = 0; r < rows; ++r) s_map[r] = s_alive[r] ? cnt++ : -1;
        if (cnt > cap) { s_flag |= 2; }
        s_piv = cnt;
    }
    __syncthreads();
    const int cnt = s_piv < cap ? s_piv : cap;
    double *Ao = Ar + (size_t)t * (size_t)(n + p) * cap;
    for (int c = 0; c < n + p; ++c) {
        const int cs = c < n ? c : c + m;
        for (int r = tid; r < rows; r += 256) {
            const int o = s_map[r];
            if (o >= 0 && o < cap) Ao[(size_t)c * cap + o] = A[(size_t)cs * rows + r];
        }
        for (int o = cnt + tid; o < cap; o += 256) Ao[(size_t)c * cap + o] = 0.0;
    }
    for (int r = tid; r < rows; r += 256) {
        const int o = s_map[r];
        if (o >= 0 && o < cap) { lr[(size_t)t * cap + o] = s_l[r]; ur[(size_t)t * cap + o] = s_u[r]; }
    }
    for (int o = cnt + tid; o < cap; o += 256) { lr[(size_t)t * cap + o] = -QINF; ur[(size_t)t * cap + o] = QINF; }
    if (tid == 0) { rows_out[t] = cnt; flags_out[t] = s_flag; }
}

} // namespace

hipError_t qpn_launch_recipes_batch(int32_t nodes, int32_t N, const uint8_t *masks, const long long *offsets, long long total, uint8_t *K,
                                    int32_t *node_of, hipStream_t stream)
{
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(recipes_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, nodes, N, masks, offsets, total, K,
                       node_of);
    return hipGetLastError();
}

size_t qpn_reduce_pieces_lds(int32_t n, int32_t m, int32_t p)
{
    const size_t N = (size_t)n + m, rows = 2 * N, cols = N + p;
    return (3 * rows + cols) * sizeof(double) + rows * sizeof(int) + ((rows + 7) & ~(size_t)7);
}

hipError_t qpn_launch_reduce_pieces(int32_t pieces, int32_t n, int32_t m, int32_t p, double tol, double *Ap, const double *lp,
                                    const double *up, const uint8_t *keep, double *Ar, double *lr, double *ur, int32_t *rows_out,
                                    int32_t *flags_out, hipStream_t stream)
{
    if (pieces <= 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_pieces_kernel, dim3((unsigned)pieces), dim3(256), qpn_reduce_pieces_lds(n, m, p), stream, n, m, p, tol, Ap, lp,
                       up, keep, Ar, lr, ur, rows_out, flags_out);
    return hipGetLastError();
}
