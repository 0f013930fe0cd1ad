#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_fma(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double c[16]; for (int j = 0; j < 16; ++j) c[j] = j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = fma(a, b, c[j]);
    }
    double s = 0; for (int j = 0; j < 16; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out; hipMalloc(&out, 8 * 2048 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {
        int blocks = 1024 * waves, iters = 4096; float ms;
        k_mfma<<<blocks, 64>>>(out, 16); hipDeviceSynchronize();
        hipEventRecord(e0); k_mfma<<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * iters * 4 * 2048.0;
        printf("MFMA f64 16x16x4: %d waves (1 per block): %.3f ms -> %.1f TFLOP/s; per-wave cycles per MFMA at 2.4GHz: %.1f\n", blocks, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0) / ((blocks + 1023) / 1024));
        k_fma<<<blocks, 64>>>(out, 16); hipDeviceSynchronize();
        hipEventRecord(e0); k_fma<<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        flops = (double)blocks * iters * 16 * 64 * 2.0;
        printf("v_fma_f64        : %d waves: %.3f ms -> %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
    return 0;
}
