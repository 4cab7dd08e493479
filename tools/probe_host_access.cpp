// Probe: can host code dereference memory handed out by the allocation hook?  (the unchanged Fortran driver
// touches q, hq, ... with array syntax.)  Tries hipMalloc, fine-grained device memory and managed memory; each variant in a child process
// so that a fault in one does not hide the others.  Build: hipcc --offload-arch=gfx950 -o probe tools/probe_host_access.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

__global__ void k_scale(double *a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) a[i] = 2.0 * a[i] + 1.0;
}

static int run(int variant) {
    const size_t n = 1 << 24;   // 128 MB
    double *p = nullptr;
    hipError_t e = hipSuccess;
    const char *name = "";
    if (variant == 0) { name = "hipMalloc"; e = hipMalloc((void **)&p, n * 8); }
    if (variant == 1) { name = "hipExtMallocWithFlags(finegrained)"; e = hipExtMallocWithFlags((void **)&p, n * 8, hipDeviceMallocFinegrained); }
    if (variant == 2) { name = "hipMallocManaged"; e = hipMallocManaged((void **)&p, n * 8); }
    if (variant == 3) {
        name = "hipMallocManaged+advise(device)";
        e = hipMallocManaged((void **)&p, n * 8);
        if (e == hipSuccess) { hipMemAdvise(p, n * 8, hipMemAdviseSetPreferredLocation, 0); hipMemPrefetchAsync(p, n * 8, 0, 0); hipDeviceSynchronize(); }
    }
    if (e != hipSuccess) { printf("%-40s alloc failed: %s\n", name, hipGetErrorString(e)); return 1; }
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; ++i) p[i] = (double)i;            // host write
    auto t1 = std::chrono::steady_clock::now();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_scale<<<(unsigned)((n + 255) / 256), 256>>>(p, n); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) k_scale<<<(unsigned)((n + 255) / 256), 256>>>(p, n);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    auto t2 = std::chrono::steady_clock::now();
    double s = 0; for (size_t i = 0; i < n; i += 1) s += p[i];  // host read
    auto t3 = std::chrono::steady_clock::now();
    auto sec = [](auto x, auto y) { return std::chrono::duration<double>(y - x).count(); };
    printf("%-40s host write %.2f GB/s, host read %.2f GB/s, kernel %.1f GB/s (read+write), sum %.6e\n", name, n * 8 / sec(t0, t1) / 1e9,
           n * 8 / sec(t2, t3) / 1e9, 10.0 * 2 * n * 8 / (ms * 1e-3) / 1e9, s);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1) return run(atoi(argv[1]));
    for (int v = 0; v < 4; ++v) {
        fflush(stdout);
        pid_t pid = fork();                 // the parent has not touched the GPU
        if (pid == 0) { execl(argv[0], argv[0], (v == 0 ? "0" : v == 1 ? "1" : v == 2 ? "2" : "3"), (char *)nullptr); _exit(127); }
        int st = 0; waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) printf("variant %d: killed by signal %d (host access faults)\n", v, WTERMSIG(st));
        else if (WEXITSTATUS(st) != 0) printf("variant %d: exit %d\n", v, WEXITSTATUS(st));
    }
    return 0;
}
