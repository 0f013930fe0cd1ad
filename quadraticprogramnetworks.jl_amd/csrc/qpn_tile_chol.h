// Blocked Cholesky of a symmetric positive definite matrix in LDS, for one workgroup of 256 threads (four wavefronts).
// The matrix is the packed LOWER triangle of 16 x 16 tiles (tile (i, j), j <= i, at tc_toff(i, j); row stride 17 doubles: the
// operand reads of v_mfma_f64_16x16x4_f64 -- element [lc][4 s + lq] of a tile for both operands of a product -- fall on
// distinct banks), k rows padded to T = ceil(k / 16) tiles with the identity.  Right-looking: per tile column j, wave 0 factors
// the diagonal tile in registers (lane <-> row, pivots and columns broadcast by v_readlane, v_rsq_f64 + three Newton steps instead
// of sqrt and division) and leaves the tile's INVERSE in its place; the panel L(i, j) = A(i, j) L_jj^-T and the trailing update
// A(i, l) -= L(i, j) L(l, j)' run on the matrix cores, one tile per wave at a time -- with a look-ahead: the next diagonal tile is
// updated first and factored by wave 0 while waves 1 .. 3 finish the update.  Users: schur_big_bpp (qpn_avi_schur_big.hip),
// verify_wide_node (qpn_verify.hip).
#pragma once
#include "qpn_internal.h"

constexpr int TC_TLD = 17, TC_TSZ = 16 * TC_TLD;
__host__ __device__ constexpr int tc_tiles(int T) { return T * (T + 1) / 2; }
__device__ __forceinline__ int tc_toff(int i, int j) { return (i * (i + 1) / 2 + j) * TC_TSZ; }
__device__ __forceinline__ void tc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Every thread of the workgroup calls it (block barriers inside).  *fail (LDS, 0 on entry) is set when a pivot is not above
// piv_rel x max(1, first diagonal of its tile); returns false then (uniform), the tiles are left half-factored.
__device__ __forceinline__ bool tc_factor(double *tiles, int T, int *fail, double piv_rel, int tid)
{
    typedef double tc_d4 __attribute__((ext_vector_type(4)));
    const int wave = tid >> 6, lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    // the diagonal tile's turn on ONE wavefront: factor in registers, inverse back in place
    auto diag_tile = [&](int j) {
        double *const D = tiles + tc_toff(j, j);
        const int r = lane & 15;
        double av[16], dinv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) av[c] = D[r * TC_TLD + c];
        const double d0 = fabs(readlane_f64(av[0], 0));
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double pv = readlane_f64(av[c], c);
            if (!(pv > piv_rel * fmax(d0, 1.0))) bad = true;
            // 1 / sqrt(pv): the hardware estimate and three Newton steps (no fp64 sqrt / division on the serial chain)
            const double pq = bad ? 1.0 : pv;
            double rs = __builtin_amdgcn_rsq(pq);
#pragma unroll
            for (int nw = 0; nw < 3; ++nw) { const double e_ = fma(-pq * rs, rs, 1.0); rs = fma(0.5 * rs, e_, rs); }
            dinv[c] = rs;
            const double lcol = av[c] * rs;
            av[c] = lcol;
#pragma unroll
            for (int c2 = c + 1; c2 < 16; ++c2) av[c2] = fma(-lcol, readlane_f64(lcol, c2), av[c2]);
        }
        // column `r` of the inverse: x_i = (delta_ir - sum_{t < i} L_it x_t) / L_ii
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double acc = i == r ? 1.0 : 0.0;
#pragma unroll
            for (int t = 0; t < i; ++t) acc = fma(-readlane_f64(av[t], i), x[t], acc);
            x[i] = acc * dinv[i];
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) D[i * TC_TLD + r] = x[i];
        }
        if (bad && lane == 0) *fail = 1;
    };
    auto update_tile = [&](int j, int i, int l2) {               // A(i, l2) -= L(i, j) L(l2, j)'
        const double *const Lij = tiles + tc_toff(i, j), *const Llj = tiles + tc_toff(l2, j);
        double *const C = tiles + tc_toff(i, l2);
        tc_d4 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = C[(4 * g + lq) * TC_TLD + lc];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lij[lc * TC_TLD + 4 * s4 + lq], Llj[lc * TC_TLD + 4 * s4 + lq], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) C[(4 * g + lq) * TC_TLD + lc] = acc[g];
    };
    if (wave == 0) diag_tile(0);
    __syncthreads();
    if (*fail) return false;                                      // (uniform)
    for (int j = 0; j < T; ++j) {
        // panel: L(i, j) = A(i, j) Linv' -- one tile per wave at a time
        {
            const double *const Li = tiles + tc_toff(j, j);
            for (int i = j + 1 + wave; i < T; i += 4) {
                double *const Aij = tiles + tc_toff(i, j);
                tc_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Aij[lc * TC_TLD + 4 * s4 + lq], Li[lc * TC_TLD + 4 * s4 + lq], acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) Aij[(4 * g + lq) * TC_TLD + lc] = acc[g];
            }
        }
        __syncthreads();
        if (j + 1 >= T) break;
        // trailing update A(i, l) -= L(i, j) L(l, j)' for j < l <= i, with a look-ahead: wave 0 updates the NEXT diagonal tile
        // first and factors it at once, while waves 1 .. 3 share the rest of the update (the diagonal tile's serial chain is the
        // longest item of a step; it now runs next to the update instead of after it)
        if (wave == 0) {
            update_tile(j, j + 1, j + 1);
            tc_wave_sync();
            diag_tile(j + 1);
        } else {
            const int nt = T - 1 - j, np_ = nt * (nt + 1) / 2;
            for (int idx = wave; idx < np_; idx += 3) {           // pairs 1 .. np_ - 1 (pair 0 is wave 0's): waves 1, 2, 3
                int ii = 0, rem = idx;
                while (rem > ii) { rem -= ii + 1; ii++; }         // idx = ii (ii + 1) / 2 + rem, rem <= ii
                update_tile(j, j + 1 + ii, j + 1 + rem);
            }
        }
        __syncthreads();
        if (*fail) return false;                                  // (uniform)
    }
    return true;
}

// L y = b, L' x = y in place on rhs[0 .. 16 T) (LDS): ONE wavefront calls it (two rows per lane: T <= 8); the caller puts a block
// barrier behind it.  The diagonal tiles hold the inverses tc_factor left.
__device__ __forceinline__ void tc_solve(const double *tiles, double *rhs, int T, int lane)
{
    const int kp = 16 * T;
    for (int j = 0; j < T; ++j) {
        const double *const Li = tiles + tc_toff(j, j);
        double yv = 0.0;
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) yv = fma(Li[lane * TC_TLD + c], rhs[16 * j + c], yv);
        }
        tc_wave_sync();
        if (lane < 16) rhs[16 * j + lane] = yv;
        tc_wave_sync();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = lane + 64 * h;
            if (row >= 16 * (j + 1) && row < kp) {
                const double *const Lr = tiles + tc_toff(row >> 4, j) + (row & 15) * TC_TLD;
                double acc = rhs[row];
#pragma unroll
                for (int c = 0; c < 16; ++c) acc = fma(-Lr[c], rhs[16 * j + c], acc);
                rhs[row] = acc;
            }
        }
        tc_wave_sync();
    }
    for (int j = T - 1; j >= 0; --j) {
        // x_j = Linv_jj' r_j (lane <-> column c of tile column j: x_c = sum_{r >= c} Linv(r, c) r_r) ...
        const double *const Li = tiles + tc_toff(j, j);
        double xv = 0.0;
        if (lane < 16) {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) xv = fma(Li[r2 * TC_TLD + lane], rhs[16 * j + r2], xv);
        }
        tc_wave_sync();
        if (lane < 16) rhs[16 * j + lane] = xv;
        tc_wave_sync();
        // ... then every earlier position gives up its share: r_p -= sum_r L(16 j + r, p) x_r  (two positions per lane)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pcol = lane + 64 * h;
            if (pcol < 16 * j) {
                const double *const Lc = tiles + tc_toff(j, pcol >> 4) + (pcol & 15);
                double acc = rhs[pcol];
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) acc = fma(-Lc[r2 * TC_TLD], rhs[16 * j + r2], acc);
                rhs[pcol] = acc;
            }
        }
        tc_wave_sync();
    }
}
