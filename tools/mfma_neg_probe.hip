// Developer probe: does the BLGP field of v_mfma_f64_16x16x4_f64 act as NEG modifiers on gfx950 (bit 0: A, bit 1: B, bit 2: C)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int BLGP> __global__ void k(const double *a, const double *b, const double *c, double *d)
{
    const int l = threadIdx.x;
    d4 acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, BLGP);
    for (int g = 0; g < 4; ++g) d[l * 4 + g] = acc[g];
}
int main()
{
    double ha[64], hb[64], hc[256], hd[8][256];
    for (int i = 0; i < 64; ++i) { ha[i] = 0.5 + i * 0.01; hb[i] = 1.0 - i * 0.02; }
    for (int i = 0; i < 256; ++i) hc[i] = 0.25 * i;
    double *a, *b, *c, *d;
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&c, sizeof hc); hipMalloc(&d, sizeof hd[0]);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
#define RUN(B) k<B><<<1, 64>>>(a, b, c, d); hipMemcpy(hd[B], d, sizeof hd[0], hipMemcpyDeviceToHost);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    // blgp=1 should equal C - A*B if it negates A: d1 - c == -(d0 - c)
    double e1 = 0, e2 = 0, e4 = 0, e3 = 0;
    for (int i = 0; i < 256; ++i) {
        const double p = hd[0][i] - hc[i];
        e1 = fmax(e1, fabs((hd[1][i] - hc[i]) + p)); e2 = fmax(e2, fabs((hd[2][i] - hc[i]) + p));
        e3 = fmax(e3, fabs((hd[3][i] - hc[i]) - p)); e4 = fmax(e4, fabs((hd[4][i] + hc[i]) - p));
    }
    printf("blgp=1 negates A: err %.3e; blgp=2 negates B: err %.3e; blgp=3 negates both: err %.3e; blgp=4 negates C: err %.3e (sample prod %.4f)\n", e1, e2, e3, e4, hd[0][5] - hc[5]);
    return 0;
}
