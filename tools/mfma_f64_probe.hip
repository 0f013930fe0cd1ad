// Probe: v_mfma_f64_16x16x4_f64 operand / result layout and issue cost on gfx950.
// Layout claim (cdna_hip_programming.md section 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15] one f64
// per lane; C/D 4 f64 per lane: col = l&15, row = (l>>4) + 4*reg.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, const double *C, double *D, unsigned long long *cyc)
{
    const int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];        // A is 16x4 row-major
    double b = B[(l >> 4) * 16 + (l & 15)];       // B is 4x16 row-major
    d4 c;
    for (int g = 0; g < 4; ++g) c[g] = C[((l >> 4) + 4 * g) * 16 + (l & 15)];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int g = 0; g < 4; ++g) D[((l >> 4) + 4 * g) * 16 + (l & 15)] = d[g];
    d4 acc0 = c, acc1 = c, acc2 = c, acc3 = c;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 16; ++it) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    d4 accd = c;
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) accd = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, accd, 0, 0, 0);
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
    D[256 + l] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + accd[0];
}
int main()
{
    double hA[64], hB[64], hC[256], hD[512], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (i + 1) + 0.25 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (j * 3 + 1) - 0.5 * k * k;
    for (int i = 0; i < 256; ++i) hC[i] = 0.001 * i;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = hC[i * 16 + j];
        for (int k = 0; k < 4; ++k) s = fma(hA[i * 4 + k], hB[k * 16 + j], s);
        ref[i * 16 + j] = s;
    }
    double *dA, *dB, *dC, *dD; unsigned long long *dcyc, hc[2];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD); hipMalloc(&dcyc, 16);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, dD, dcyc);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hc, dcyc, 16, hipMemcpyDeviceToHost);
    double err = 0; int exact = 1;
    for (int i = 0; i < 256; ++i) { err = fmax(err, fabs(hD[i] - ref[i])); if (hD[i] != ref[i]) exact = 0; }
    printf("layout check: max |D - ref| = %g, bitwise equal to k-ordered fma chain: %s\n", err, exact ? "yes" : "no");
    printf("64 independent MFMAs (4 accumulators): %llu cycles (%.1f per MFMA); 64 dependent: %llu cycles (%.1f per MFMA)\n",
           hc[0], hc[0] / 64.0, hc[1], hc[1] / 64.0);
    return 0;
}
