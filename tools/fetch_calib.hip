// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the AVI kernels use
// (guide: MI355X_MICROARCH.md, section HBM -- only 16 B/lane streaming is pre-calibrated; other
// widths must be calibrated on a known byte count in the same access pattern).
// Reads a 1 GiB buffer (>> 256 MiB Infinity Cache) once with 8 B/lane coalesced loads
// (global_load_dwordx2, 512 B per wave instruction: the M-block load pattern) and once with
// 16 B/lane loads; writes 64 MiB with 8 B/lane stores.  Known bytes are printed; compare with
// the counters of the matching kernel names.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void calib_read8(const double *p, size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) s += p[i];
    if (s == 123.456) out[0] = s;
}
__global__ void calib_read16(const double2 *p, size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ void calib_write8(double *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1.0;
}
int main()
{
    const size_t bytes = 1ull << 30, wbytes = 64ull << 20;
    double *buf, *out, *wb;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8); hipMalloc(&wb, wbytes);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    calib_read8<<<2048, 256>>>(buf, bytes / 8, out);
    calib_read16<<<2048, 256>>>((const double2 *)buf, bytes / 16, out);
    calib_write8<<<2048, 256>>>(wb, wbytes / 8);
    hipDeviceSynchronize();
    printf("calib_read8 known_bytes %zu\ncalib_read16 known_bytes %zu\ncalib_write8 known_bytes %zu\n", bytes, bytes, wbytes);
    return 0;
}
