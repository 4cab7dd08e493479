// Does the speed of a many-stream kernel depend on where hipMalloc put its arrays?  (profiles/r05/sustained_probe.txt: the same binary runs the
// bandwidth-bound kernels 9 % slower in one process than in the next -- per process, not over time.)
// 13 arrays of 1 GiB (8 read, 5 written per point, as the fused x-Burgers launch), allocated (a) by hipMalloc as they come, (b) by hipMalloc after a
// few small allocations that shift the addresses, (c) through the virtual-memory API with the virtual address aligned to `align` bytes.
// Prints, per set: the alignment of every array's address (log2) and the rate of the streaming kernel over the set.
//     hipcc -O3 --offload-arch=gfx950 tools/placement_probe.hip -o tools/placement_probe && tools/placement_probe
#include <hip/hip_runtime.h>

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

__global__ void __launch_bounds__(256) k_copy(const double2 *a, double2 *b, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

__global__ void __launch_bounds__(256) k_read(const double2 *a, double *out, size_t n2) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(double2 *a, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) a[i] = make_double2(1.0, 2.0);
}

template <class F>
static double timed(F launch, int reps, double bytes_per_launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return bytes_per_launch * reps / (ms * 1e-3) * 1e-9;
}

static int ctz64(uintptr_t v) { return v ? __builtin_ctzll(v) : 64; }

static double time_stream(const std::vector<void *> &a, size_t bytes) {
    Ptrs p;
    for (int k = 0; k < 8; ++k) p.r[k] = (const double2 *)a[k];
    for (int k = 0; k < 5; ++k) p.w[k] = (double2 *)a[8 + k];
    const size_t n2 = bytes / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_stream13, dim3(8192), dim3(256), 0, 0, p, n2);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k_stream13, dim3(8192), dim3(256), 0, 0, p, n2);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 13.0 * bytes * reps / (ms * 1e-3) * 1e-9;
}

static double time_copy(void *a, void *b, size_t bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, bytes / 16);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, bytes / 16);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * bytes * 5 / (ms * 1e-3) * 1e-9;
}

static void report(const char *what, const std::vector<void *> &a, size_t bytes) {
    printf("%-44s log2(alignment of the 13 addresses):", what);
    for (void *p : a) printf(" %d", ctz64((uintptr_t)p));
    const double g = time_stream(a, bytes);
    printf("   13-stream %.0f GB/s   copy a0->a8 %.0f GB/s\n", g, time_copy(a[0], a[8], bytes));
    fflush(stdout);
}

static void survey(int N, size_t bytes) {
    // per array: read alone, written alone; then 13-stream sets drawn from the pool: does the rate of a set follow from its members' own rates?
    std::vector<void *> a(N);
    for (auto &p : a) CK(hipMalloc(&p, bytes));
    double *out;
    CK(hipMalloc((void **)&out, 64));
    std::vector<double> rd(N), wr(N);
    for (int i = 0; i < N; ++i) {
        rd[i] = timed([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)a[i], out, bytes / 16); }, 5, (double)bytes);
        wr[i] = timed([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (double2 *)a[i], bytes / 16); }, 5, (double)bytes);
    }
    printf("survey of %d arrays of %zu MiB (hipMalloc, back to back): read alone / written alone [GB/s]\n", N, bytes >> 20);
    for (int i = 0; i < N; ++i) printf("  array %2d  read %5.0f  write %5.0f\n", i, rd[i], wr[i]);
    unsigned seed = 12345;
    auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int t = 0; t < 12; ++t) {
        std::vector<int> idx(N);
        for (int i = 0; i < N; ++i) idx[i] = i;
        for (int i = N - 1; i > 0; --i) std::swap(idx[i], idx[rnd() % (i + 1)]);
        std::vector<void *> set(13);
        double mr = 0, mw = 0;
        for (int k = 0; k < 13; ++k) { set[k] = a[idx[k]]; if (k < 8) mr += rd[idx[k]] / 8; else mw += wr[idx[k]] / 5; }
        const double g = time_stream(set, bytes), g2 = time_stream(set, bytes);
        printf("  set");
        for (int k = 0; k < 13; ++k) printf(" %2d", idx[k]);
        printf("   13-stream %5.0f, again %5.0f GB/s   (members alone: read %5.0f, write %5.0f)\n", g, g2, mr, mw);
    }
    // structured sets: consecutive allocations, every 2nd / 3rd, and pairs at a given distance (copy)
    auto strided = [&](int start, int step, const char *what) {
        std::vector<void *> set(13);
        for (int k = 0; k < 13; ++k) set[k] = a[(start + k * step) % N];
        printf("  %-40s 13-stream %5.0f GB/s\n", what, time_stream(set, bytes));
    };
    strided(0, 1, "arrays 0..12 (consecutive allocations)");
    strided(13, 1, "arrays 13..25");
    strided(26, 1, "arrays 26..38");
    strided(0, 2, "every 2nd array from 0");
    strided(0, 3, "every 3rd array from 0");
    strided(1, 3, "every 3rd array from 1");
    for (int step : {1, 2, 3, 4, 5, 6, 7, 8})
        for (int start : {0, 1, 2}) {
            if (start + 12 * step >= N) continue;
            char nm[64];
            snprintf(nm, sizeof nm, "every %d-th array from %d", step, start);
            strided(start, step, nm);
        }
    {   // interleave reads and writes differently: reads 0..7 from the low arrays, writes to far-away ones
        std::vector<void *> set(13);
        for (int k = 0; k < 8; ++k) set[k] = a[k];
        for (int k = 0; k < 5; ++k) set[8 + k] = a[N - 1 - k];
        printf("  %-40s 13-stream %5.0f GB/s\n", "reads 0..7, writes N-1..N-5", time_stream(set, bytes));
    }
    for (int dist : {1, 2, 3, 4, 5, 8, 13, 21})
        printf("  copy array 0 -> array %2d: %5.0f GB/s     copy array 7 -> array %2d: %5.0f GB/s\n", dist, time_copy(a[0], a[dist], bytes), 7 + dist,
               time_copy(a[7], a[7 + dist], bytes));
    for (auto p : a) CK(hipFree(p));
    CK(hipFree(out));
    // one allocation, arrays carved out of it `skew` bytes further apart than their size
    for (size_t skew : {(size_t)0, (size_t)256, (size_t)4096, (size_t)4352, (size_t)65536, (size_t)(1u << 20), (size_t)(2u << 20), (size_t)(3u << 20) + 4352, (size_t)(32u << 20),
                        (size_t)(100u << 20) + 8448}) {
        char *arena;
        if (hipMalloc((void **)&arena, 13 * (bytes + skew)) != hipSuccess) { printf("arena allocation failed\n"); break; }
        std::vector<void *> set(13);
        for (int k = 0; k < 13; ++k) set[k] = arena + k * (bytes + skew);
        printf("  one allocation, arrays %zu + %zu bytes apart: 13-stream %5.0f GB/s\n", bytes, skew, time_stream(set, bytes));
        CK(hipFree(arena));
    }
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)1 << 30;
    CK(hipSetDevice(0));
    if (argc > 1 && atoi(argv[1]) > 0) { survey(atoi(argv[1]), bytes); return 0; }
    {   // (a) as they come
        std::vector<void *> a(13);
        for (auto &p : a) CK(hipMalloc(&p, bytes));
        report("hipMalloc, back to back", a, bytes);
        for (auto p : a) CK(hipFree(p));
    }
    for (int shift = 1; shift <= 3; ++shift) {   // (b) small allocations in between
        std::vector<void *> a(13), junk;
        for (auto &p : a) {
            void *j;
            CK(hipMalloc(&j, (size_t)shift * (2u << 20) + (6u << 20)));
            junk.push_back(j);
            CK(hipMalloc(&p, bytes));
        }
        char name[80];
        snprintf(name, sizeof name, "hipMalloc, %d MB allocations in between", shift * 2 + 6);
        report(name, a, bytes);
        for (auto p : a) CK(hipFree(p));
        for (auto p : junk) CK(hipFree(p));
    }
    {   // (a') sizes that are not a power of two (the txc arrays of the box: (nx + 2) ny nz doubles)
        const size_t b2 = (size_t)514 * 512 * 512 * 8;
        std::vector<void *> a(13);
        for (auto &p : a) CK(hipMalloc(&p, b2));
        report("hipMalloc of 514 x 512 x 512 doubles", a, bytes);
        for (auto p : a) CK(hipFree(p));
    }
    // (c) virtual-memory API: address reserved with a chosen alignment, physical memory created and mapped
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("no virtual-memory API\n"); return 0; }
    printf("virtual-memory API: recommended granularity %zu bytes\n", gran);
    for (int lg : {21, 25, 30}) {
        std::vector<void *> a(13);
        std::vector<hipMemGenericAllocationHandle_t> h(13);
        bool okay = true;
        for (int k = 0; k < 13 && okay; ++k) {
            if (hipMemCreate(&h[k], bytes, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); okay = false; break; }
            if (hipMemAddressReserve(&a[k], bytes, (size_t)1 << lg, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve(align 2^%d) failed\n", lg); okay = false; break; }
            CK(hipMemMap(a[k], bytes, 0, h[k], 0));
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(a[k], bytes, &acc, 1));
        }
        if (!okay) continue;
        char name[80];
        snprintf(name, sizeof name, "hipMemCreate + address aligned to 2^%d", lg);
        report(name, a, bytes);
        for (int k = 0; k < 13; ++k) { CK(hipMemUnmap(a[k], bytes)); CK(hipMemAddressFree(a[k], bytes)); CK(hipMemRelease(h[k])); }
    }
    // (d) ONE virtual range for all 13 arrays (what a host's 2-D array q(isize_field, 3) is) backed by 13 physical allocations of their own:
    // created one after the other / with throw-away allocations of irregular sizes created in between (released afterwards) / in shuffled order
    for (int variant = 0; variant < 6; ++variant) {
        std::vector<hipMemGenericAllocationHandle_t> h(13), junk;
        std::vector<void *> a(13);
        void *base = nullptr;
        if (hipMemAddressReserve(&base, 13 * bytes, 0, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve of 13 arrays failed\n"); break; }
        unsigned seed = 777u + (unsigned)variant;
        auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 10; };
        bool okay = true;
        for (int k = 0; k < 13 && okay; ++k) {
            if (variant >= 1 && variant <= 3) {      // a throw-away allocation of 2 .. 510 MiB in front of every array
                hipMemGenericAllocationHandle_t j;
                const size_t jb = ((size_t)(1 + rnd() % 255)) << 21;
                if (hipMemCreate(&j, jb, &prop, 0) == hipSuccess) junk.push_back(j);
            }
            okay = hipMemCreate(&h[k], bytes, &prop, 0) == hipSuccess;
        }
        if (!okay) { printf("hipMemCreate failed\n"); break; }
        std::vector<int> order(13);
        for (int k = 0; k < 13; ++k) order[k] = k;
        if (variant >= 4)
            for (int k = 12; k > 0; --k) std::swap(order[k], order[rnd() % (k + 1)]);
        for (int k = 0; k < 13; ++k) {
            a[k] = (char *)base + (size_t)k * bytes;
            CK(hipMemMap(a[k], bytes, 0, h[order[k]], 0));
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(base, 13 * bytes, &acc, 1));
        const char *names[6] = {"one range, 13 physical allocations in order", "one range, irregular throw-aways in between (1)", "one range, irregular throw-aways in between (2)",
                                "one range, irregular throw-aways in between (3)", "one range, physical allocations shuffled (1)", "one range, physical allocations shuffled (2)"};
        report(names[variant], a, bytes);
        for (int k = 0; k < 13; ++k) CK(hipMemUnmap(a[k], bytes));
        CK(hipMemAddressFree(base, 13 * bytes));
        for (auto x : h) CK(hipMemRelease(x));
        for (auto x : junk) CK(hipMemRelease(x));
    }
    // (e) the same with the arrays (1 GiB + pad) apart in the VIRTUAL range -- the physical allocations are what they are
    for (size_t padm : {(size_t)0, (size_t)1, (size_t)2, (size_t)3, (size_t)5, (size_t)8, (size_t)13, (size_t)16, (size_t)21, (size_t)32, (size_t)64, (size_t)100, (size_t)257}) {
        const size_t stride = bytes + (padm << 21);
        std::vector<hipMemGenericAllocationHandle_t> h(13);
        std::vector<void *> a(13);
        void *base = nullptr;
        if (hipMemAddressReserve(&base, 13 * stride, 0, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve failed\n"); break; }
        bool okay = true;
        for (int k = 0; k < 13 && okay; ++k) okay = hipMemCreate(&h[k], bytes, &prop, 0) == hipSuccess;
        if (!okay) { printf("hipMemCreate failed\n"); break; }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (int k = 0; k < 13; ++k) {
            a[k] = (char *)base + (size_t)k * stride;
            CK(hipMemMap(a[k], bytes, 0, h[k], 0));
            CK(hipMemSetAccess(a[k], bytes, &acc, 1));
        }
        char name[96];
        snprintf(name, sizeof name, "one range, arrays 1 GiB + %zu x 2 MiB apart", padm);
        report(name, a, bytes);
        for (int k = 0; k < 13; ++k) CK(hipMemUnmap(a[k], bytes));
        CK(hipMemAddressFree(base, 13 * stride));
        for (auto x : h) CK(hipMemRelease(x));
    }
    {   // one hipMalloc for all 13
        char *arena;
        if (hipMalloc((void **)&arena, 13 * bytes) == hipSuccess) {
            std::vector<void *> a(13);
            for (int k = 0; k < 13; ++k) a[k] = arena + (size_t)k * bytes;
            report("one hipMalloc for all 13 arrays", a, bytes);
            CK(hipFree(arena));
        }
    }
    {   // (a) again, at the end: has the state of the process changed?
        std::vector<void *> a(13);
        for (auto &p : a) CK(hipMalloc(&p, bytes));
        report("hipMalloc, back to back (again)", a, bytes);
        for (auto p : a) CK(hipFree(p));
    }
    return 0;
}
