// Does a wave's fp64 MFMA stream overlap with fp64 VALU work on the same SIMD (gfx950)?
// (a) one wave: MFMA only / FMA only / both interleaved;  (b) two waves per SIMD: one MFMA-only, one FMA-only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void k(double *out, unsigned long long *cyc, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    double c[16]; for (int j = 0; j < 16; ++j) c[j] = j;
    // MODE 3: even blocks MFMA only, odd blocks FMA only (two waves per SIMD when grid = 2048)
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1) == 0);
    const bool do_f = MODE == 1 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1) == 1);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (do_m) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); }
        if (do_f) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j] = fma(a, b, c[j]);
        }
        if (do_m) { c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0); }
        if (do_f) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j] = fma(b, a, c[j]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = c0[0] + c1[1]; for (int j = 0; j < 16; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h[4096];
    hipMalloc(&out, 8 * 4096 * 64); hipMalloc(&cyc, 8 * 4096);
    const int iters = 2048;
    auto run = [&](int mode, int blocks, const char *nm) {
        switch (mode) { case 0: k<0><<<blocks, 64>>>(out, cyc, iters); break; case 1: k<1><<<blocks, 64>>>(out, cyc, iters); break;
                        case 2: k<2><<<blocks, 64>>>(out, cyc, iters); break; default: k<3><<<blocks, 64>>>(out, cyc, iters); }
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double e = 0, o = 0; for (int i = 0; i < blocks; i += 2) { e += h[i]; o += h[i + 1]; }
        printf("%-46s blocks %4d: cycles/iter (2 MFMA and/or 32 FMA): even blocks %.1f, odd blocks %.1f\n", nm, blocks,
               e / (blocks / 2) / iters, o / (blocks / 2) / iters);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run(0, 1024, "MFMA only, 1 wave/SIMD");
        run(1, 1024, "FMA only, 1 wave/SIMD");
        run(2, 1024, "MFMA+FMA interleaved in one wave, 1 wave/SIMD");
        run(3, 2048, "2 waves/SIMD: even=MFMA only, odd=FMA only");
        run(0, 2048, "MFMA only, 2 waves/SIMD");
        run(1, 2048, "FMA only, 2 waves/SIMD");
    }
    return 0;
}
