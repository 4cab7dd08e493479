// Streaming yardsticks for one MI355X: what plain hand-written kernels reach on THIS box, so that the Burgers / derivative kernels are read against
// the memory system's own ceiling for THEIR access pattern and not against torch.copy_.
//   hipcc -O3 --offload-arch=gfx950 tools/yardstick.hip -o tools/yardstick && tools/yardstick [n]
// Lines printed (JSON, one per kernel): bytes moved / median launch time (HIP events, 10 launches after 2 warm-ups).
//   copy16 / triad16      global_load_dwordx4 / global_store_dwordx4, grid-stride, 1 read + 1 write / 3 reads + 1 write
//   copy8                 the same copy with 8 B per lane (512 B per wave instruction): the width of the tile kernels' accesses
//   tile<L,dir,phased>    the access pattern of k_htile / k_rtile WITHOUT arithmetic: a 16-wave (L = 64) or 8-wave (L = 32) workgroup owns L
//                         memory-contiguous lines x 512 rows (row stride nx for y lines, nx*ny for z lines), reads operand + velocity + old tendency,
//                         writes the new tendency (32 B per point); phased = barriers between the loads as the solves put them
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void __launch_bounds__(256) copy16(const double2 *__restrict__ a, double2 *__restrict__ o, long long n2) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        o[i] = v0; o[i + stride] = v1; o[i + 2 * stride] = v2; o[i + 3 * stride] = v3;
    }
    for (; i < n2; i += stride) o[i] = a[i];
}
__global__ void __launch_bounds__(256) copy8(const double *__restrict__ a, double *__restrict__ o, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[i + k * stride] = v[k];
    }
    for (; i < n; i += stride) o[i] = a[i];
}
__global__ void __launch_bounds__(256) triad16(const double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c,
                                               double2 *__restrict__ o, long long n2) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n2; i += 2 * stride) {
        const double2 a0 = a[i], a1 = a[i + stride], b0 = b[i], b1 = b[i + stride], c0 = c[i], c1 = c[i + stride];
        o[i] = make_double2(a0.x + b0.x * c0.x, a0.y + b0.y * c0.y);
        o[i + stride] = make_double2(a1.x + b1.x * c1.x, a1.y + b1.y * c1.y);
    }
    for (; i < n2; i += stride) o[i] = make_double2(a[i].x + b[i].x * c[i].x, a[i].y + b[i].y * c[i].y);
}
// read-only sweep (sum kept so that the loads stay): the read ceiling alone
__global__ void __launch_bounds__(256) read16(const double2 *__restrict__ a, double *__restrict__ sink, long long n2) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    if (s == 1.2345e300) sink[0] = s;
}

// L lines x (C * M) rows per workgroup, thread (l, c) owns rows [c M, (c+1) M) of line l
// P1 = true: the first-derivative pattern (read the operand, write the result: 16 B per point); false: the Burgers pattern (32 B per point)
template <int L, int M, bool PHASED, bool NT, bool P1 = false>
__global__ void __launch_bounds__(L * 16) tile(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ o, int lines_inner,
                                               long long outer_stride, long long rs, int xcd_map) {
    const int l = threadIdx.x % L, c = threadIdx.x / L;
    const int tiles_inner = lines_inner / L;
    long long t = blockIdx.x;
    const long long outer = t / tiles_inner;
    const long long base = outer * outer_stride + (t % tiles_inner) * L + l + (long long)c * M * rs;
    double e[M], v[M], h[M];
#pragma unroll
    for (int p = 0; p < M; ++p) e[p] = a[base + p * rs];
    if (PHASED) __syncthreads();
    if constexpr (P1) {
#pragma unroll
        for (int p = 0; p < M; ++p) o[base + p * rs] = e[p] * 1.5;
        return;
    }
#pragma unroll
    for (int p = 0; p < M; ++p) v[p] = b[base + p * rs];
    if (PHASED) {
        // stand-in for the two solves: the loads above are waited for here
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < M; ++p) s += e[p];
        if (s == 1.2345e300) o[0] = s;
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < M; ++p) h[p] = NT ? __builtin_nontemporal_load(&o[base + p * rs]) : o[base + p * rs];
#pragma unroll
    for (int p = 0; p < M; ++p) {
        const double r = h[p] + e[p] * v[p];
        if (NT) __builtin_nontemporal_store(r, &o[base + p * rs]);
        else o[base + p * rs] = r;
    }
}

template <class F>
static double time_ms(F f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    std::vector<float> ms;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[5];
}

static void line(const char *name, double bytes, double ms) {
    printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"GBps\": %.1f, \"bytes\": %.0f}\n", name, ms, bytes / (ms * 1e-3) / 1e9, bytes);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const long long N = (long long)n * n * n;
    double *a, *b, *c, *o;
    CK(hipMalloc(&a, N * 8));
    CK(hipMalloc(&b, N * 8));
    CK(hipMalloc(&c, N * 8));
    CK(hipMalloc(&o, N * 8));
    CK(hipMemset(a, 0, N * 8));
    CK(hipMemset(b, 0, N * 8));
    CK(hipMemset(c, 0, N * 8));
    CK(hipMemset(o, 0, N * 8));
    for (int wg : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy16 grid=%d", wg);
        line(nm, 16.0 * N, time_ms([&] { hipLaunchKernelGGL(copy16, dim3(wg), dim3(256), 0, 0, (const double2 *)a, (double2 *)o, N / 2); }));
    }
    line("copy8 grid=8192", 16.0 * N, time_ms([&] { hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, a, o, N); }));
    line("read16 grid=8192", 8.0 * N, time_ms([&] { hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const double2 *)a, o, N / 2); }));
    line("triad16 grid=8192", 32.0 * N,
         time_ms([&] { hipLaunchKernelGGL(triad16, dim3(8192), dim3(256), 0, 0, (const double2 *)a, (const double2 *)b, (const double2 *)c, (double2 *)o, N / 2); }));
    line("hipMemcpyDtoD", 16.0 * N, time_ms([&] { CK(hipMemcpyAsync(o, a, N * 8, hipMemcpyDeviceToDevice, 0)); }));
    if (n % 512 == 0) {
        // y lines: lines_inner = nx, row stride nx, outer = z with stride nx*ny (only for n = 512: M * 16 rows = n)
        const long long nxy = (long long)n * n;
#define TILE(L, PH, NT, NAME, LI, OS, RS)                                                                                            \
    line(NAME, 32.0 * N, time_ms([&] {                                                                                               \
             hipLaunchKernelGGL((tile<L, 32, PH, NT>), dim3((unsigned)(N / (L * 512))), dim3(L * 16), 0, 0, a, b, o, (int)(LI), (long long)(OS), \
                                (long long)(RS), 0);                                                                                 \
         }))
#define TILE1(L, PH, NAME, LI, OS, RS)                                                                                               \
    line(NAME, 16.0 * N, time_ms([&] {                                                                                               \
             hipLaunchKernelGGL((tile<L, 32, PH, false, true>), dim3((unsigned)(N / (L * 512))), dim3(L * 16), 0, 0, a, b, o, (int)(LI),         \
                                (long long)(OS), (long long)(RS), 0);                                                                \
         }))
        if (n == 512) {
            TILE(32, false, false, "tile L=32 y-lines", n, nxy, n);
            TILE(32, true, false, "tile L=32 y-lines phased", n, nxy, n);
            TILE(32, true, true, "tile L=32 y-lines phased nt", n, nxy, n);
            TILE1(32, false, "tile P1 L=32 y-lines", n, nxy, n);
            TILE1(64, false, "tile P1 L=64 y-lines", n, nxy, n);
            TILE1(64, true, "tile P1 L=64 y-lines phased", n, nxy, n);
            TILE(32, false, false, "tile L=32 z-lines", nxy, 0, nxy);
            TILE(32, true, false, "tile L=32 z-lines phased", nxy, 0, nxy);
            TILE(32, true, true, "tile L=32 z-lines phased nt", nxy, 0, nxy);
            TILE1(32, false, "tile P1 L=32 z-lines", nxy, 0, nxy);
            TILE1(64, false, "tile P1 L=64 z-lines", nxy, 0, nxy);
            TILE1(64, true, "tile P1 L=64 z-lines phased", nxy, 0, nxy);
        }
    }
    return 0;
}
