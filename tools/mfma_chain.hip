// Throughput of DEPENDENT fp64 MFMA chains with B operands from LDS, 16 wavefronts per workgroup, one workgroup per CU
// (the shape of the update loop of csrc/qpn_avi_schur_big2.hip).  Prints clocks per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(1024) void k(double *out, unsigned long long *cyc, int iters)
{
    __shared__ double sv[16 * 256];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 256; i += 1024) sv[i] = 1e-3 * i;
    __syncthreads();
    double ua[16];
    for (int s = 0; s < 16; ++s) ua[s] = 1e-3 * (threadIdx.x + s);
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // one accumulator after the other, operands from LDS
#pragma unroll
            for (int s = 0; s < 16; ++s) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], sv[((s >> 2) * 2 + 0) * 256 + (s & 3) * 64 + lane], c0, 0, 0, 1);
#pragma unroll
            for (int s = 0; s < 16; ++s) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], sv[((s >> 2) * 2 + 1) * 256 + (s & 3) * 64 + lane], c1, 0, 0, 1);
        } else if (MODE == 1) {     // two accumulators interleaved
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], sv[((s >> 2) * 2 + 0) * 256 + (s & 3) * 64 + lane], c0, 0, 0, 1);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], sv[((s >> 2) * 2 + 1) * 256 + (s & 3) * 64 + lane], c1, 0, 0, 1);
            }
        } else {                    // operands in registers
#pragma unroll
            for (int s = 0; s < 16; ++s) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], ua[15 - s], c0, 0, 0, 1);
#pragma unroll
            for (int s = 0; s < 16; ++s) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], ua[15 - s], c1, 0, 0, 1);
        }
        if (MODE == 3) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h[256];
    hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&cyc, 8 * 256);
    const int iters = 512;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (mode) { case 0: k<0><<<256, 1024>>>(out, cyc, iters); break; case 1: k<1><<<256, 1024>>>(out, cyc, iters); break;
                            case 2: k<2><<<256, 1024>>>(out, cyc, iters); break; default: k<3><<<256, 1024>>>(out, cyc, iters); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost);
            double c = 0; for (int i = 0; i < 256; ++i) c += h[i];
            // per SIMD: 4 waves x 32 MFMAs per iteration
            printf("mode %d: %.3f ms, %.1f clocks (s_memtime) per MFMA per SIMD, %.1f TFLOP/s\n", mode, ms, c / 256 / iters / 128.0,
                   256.0 * 16 * iters * 32 * 2048.0 / ms / 1e9);
        }
    return 0;
}
