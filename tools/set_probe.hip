// Box state / placement: ONE bounded attempt with the counters.  The same 13-stream kernel (8 arrays read, 5 written, as the
// fused x-Burgers launch) over sets of 1-GiB hipMalloc allocations: which thirteen decides 4.8 .. 5.9 TB/s, reproducibly per set
// (tools/placement_probe, profiles/r05/placement_survey.txt).  This program times NSETS sets (each launched LAUNCHES times, rate printed per set with
// the addresses) in a fixed order, so that a `rocprofv3 --pmc` pass of the very same command gives the counters of the same launches in the same
// order: slow sets against fast sets in ONE process -- L2 hit / miss, read-request latency, the stall counters of the path to memory, and the
// address-translation counters (the layout effect follows the VIRTUAL addresses: profiles/r05/placement_virtual_not_physical.txt).
//     hipcc -O3 --offload-arch=gfx950 tools/set_probe.hip -o tools/set_probe && tools/set_probe [nsets] [seed]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Ptrs { const double2 *r[8]; double2 *w[5]; };

__global__ void __launch_bounds__(256) k_stream13(Ptrs p, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 a = p.r[0][i];
#pragma unroll
        for (int k = 1; k < 8; ++k) { const double2 b = p.r[k][i]; a.x += b.x; a.y += b.y; }
#pragma unroll
        for (int k = 0; k < 5; ++k) p.w[k][i] = make_double2(a.x + k, a.y - k);
    }
}

int main(int argc, char **argv) {
    const int nsets = argc > 1 ? atoi(argv[1]) : 16;
    unsigned long long x = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(argc > 2 ? atoi(argv[2]) : 1);
    auto rnd = [&](int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (unsigned long long)m); };
    const int N = 32, LAUNCHES = 3;
    const size_t bytes = (size_t)1 << 30, n2 = bytes / 16;
    std::vector<void *> a(N);
    for (int i = 0; i < N; ++i) { CK(hipMalloc(&a[i], bytes)); CK(hipMemset(a[i], 0, bytes)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# set  GB/s(median of %d launches)  indices | addresses >> 21 (2-MiB page numbers)\n", LAUNCHES);
    for (int s = 0; s < nsets; ++s) {
        std::vector<int> perm(N);
        for (int i = 0; i < N; ++i) perm[i] = i;
        if (s > 0) for (int i = N - 1; i > 0; --i) std::swap(perm[i], perm[rnd(i + 1)]);      // set 0: the allocator's order
        Ptrs p;
        for (int k = 0; k < 8; ++k) p.r[k] = (const double2 *)a[perm[k]];
        for (int k = 0; k < 5; ++k) p.w[k] = (double2 *)a[perm[8 + k]];
        double g[LAUNCHES];
        for (int it = 0; it < LAUNCHES; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream13, dim3(256 * 8), dim3(256), 0, 0, p, n2);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            g[it] = 13.0 * bytes / (ms * 1e-3) * 1e-9;
        }
        std::sort(g, g + LAUNCHES);
        printf("set %2d  %6.0f  ", s, g[LAUNCHES / 2]);
        for (int k = 0; k < 13; ++k) printf(" %2d", perm[k]);
        printf(" |");
        for (int k = 0; k < 13; ++k) printf(" %llx", (unsigned long long)((uintptr_t)a[perm[k]] >> 21));
        printf("\n");
    }
    return 0;
}
