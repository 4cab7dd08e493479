// Does a producer -> consumer chain that stays inside the 256-MiB Infinity Cache run faster than one that streams through HBM?  The same 16-B-per-lane
// copy kernel (and an in-place read-modify-write) on working sets from 16 MB to 2 GB, launched back to back so that each launch finds what the previous
// one left in the cache.  One JSON line per (kernel, working set).      hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o tools/mall_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) copy16(const double2 *__restrict__ a, double2 *__restrict__ o, long long n2) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        o[i] = v0; o[i + stride] = v1; o[i + 2 * stride] = v2; o[i + 3 * stride] = v3;
    }
    for (; i < n2; i += stride) o[i] = a[i];
}
__global__ void __launch_bounds__(256) rmw16(double2 *__restrict__ a, long long n2) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        v0.x += 1.0; v1.x += 1.0; v2.x += 1.0; v3.x += 1.0;
        a[i] = v0; a[i + stride] = v1; a[i + 2 * stride] = v2; a[i + 3 * stride] = v3;
    }
    for (; i < n2; i += stride) { double2 v = a[i]; v.x += 1.0; a[i] = v; }
}
int main() {
    const size_t maxb = (size_t)2 << 30;
    double2 *a, *b;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
    CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes_mb[] = {8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048};
    for (size_t mb : sizes_mb) {
        const long long n2 = (long long)(mb << 20) / 16;      // elements per array
        const int grid = 2048, reps = std::max<int>(10, (int)(4096 / mb));
        for (int kind = 0; kind < 3; ++kind) {      // 0: a -> b (working set 2 mb), 1: ping-pong a -> b -> a, 2: in place
            for (int w = 0; w < 3; ++w) copy16<<<grid, 256>>>(a, b, n2);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) {
                if (kind == 0) copy16<<<grid, 256>>>(a, b, n2);
                else if (kind == 1) { if (r & 1) copy16<<<grid, 256>>>(b, a, n2); else copy16<<<grid, 256>>>(a, b, n2); }
                else rmw16<<<grid, 256>>>(a, n2);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = 2.0 * (double)(mb << 20) * reps;
            printf("{\"kernel\": \"%s\", \"array_MB\": %zu, \"working_set_MB\": %zu, \"GBps\": %.0f, \"us_per_launch\": %.1f}\n",
                   kind == 0 ? "copy a->b repeated" : kind == 1 ? "copy ping-pong" : "in-place rmw", mb, kind == 2 ? mb : 2 * mb, bytes / (ms * 1e-3) / 1e9, ms * 1e3 / reps);
        }
    }
    return 0;
}
