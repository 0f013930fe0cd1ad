// Accuracy of v_rcp_f64 followed by 0, 1 or 2 Newton steps, against IEEE division (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *r0, double *r1, double *r2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = fma(-v, r, 1.0); r = fma(r, e, r); r1[i] = r;
    e = fma(-v, r, 1.0); r = fma(r, e, r); r2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int ex = (int)(s % 120) - 60;
        hx[i] = ldexp(1.0 + u, ex) * ((s >> 40) & 1 ? -1.0 : 1.0);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n);
    hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2);
    hipMemcpy(h0, d0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 8 * n, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        const double t = 1.0 / hx[i];
        m0 = fmax(m0, fabs(h0[i] - t) / fabs(t)); m1 = fmax(m1, fabs(h1[i] - t) / fabs(t)); m2 = fmax(m2, fabs(h2[i] - t) / fabs(t));
    }
    printf("max relative error vs 1/x: v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e (eps = %.3e)\n", m0, m1, m2, 2.220446049250313e-16);
    return 0;
}
