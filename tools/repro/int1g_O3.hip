// Reduced reproducer of the k_int1g miscompilation (tlab_amd/csrc/poisson.hip):
//     hipcc -O3 --offload-arch=gfx950 tools/repro/int1g_O3.hip -o /tmp/int1g_O3 && /tmp/int1g_O3      -> device differs from the host
//     hipcc -O0 --offload-arch=gfx950 tools/repro/int1g_O3.hip -o /tmp/int1g_O0 && /tmp/int1g_O0      -> device equals the host bit for bit
// The SAME function body (HEPTADSS-style substitution of a 7-diagonal system + the MatMul_5d right-hand side with a unit forcing, one thread per
// column, every operand re-read from memory) runs on the host and on the device; no fused multiply-adds on either side (fp contract off).
// Host replay under the sanitizers (no GPU, no HIP):   g++ -x c++ -DHOST_ONLY -O1 -g -fsanitize=address,undefined tools/repro/int1g_O3.hip -o /tmp/int1g_san && /tmp/int1g_san
#ifndef HOST_ONLY
#include <hip/hip_runtime.h>
#else
#define __host__
#define __device__
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Args {
    int n;
    long long nm;
    int unit_row;
    double bv[3];
    double *scratch, *dst, *du;
    const double *g_fac, *g_rb, *g_rt, *g_R;
};

template <int BC, int NL>
__host__ __device__ inline void body(const Args &a, long long t) {
#pragma clang fp contract(off)
    constexpr int ndi = 7, nri = 5, idl = ndi / 2 + 1, idr = nri / 2 + 1;
    const int n = a.n;
    const long long nm = a.nm;
    auto F = [&](int k, int j) { return a.g_fac[((long long)k * n + j) * nm + t]; };
    auto RB = [&](int j1, int c) { return a.g_rb[(long long)((j1 - 1) + 5 * c) * nm + t]; };
    auto RT = [&](int r, int c1) { return a.g_rt[(long long)(r + 5 * (c1 - 1)) * nm + t]; };
    auto Rr = [&](int j, int k1) { return a.g_R[j * nri + (k1 - 1)]; };
    auto fv = [&](int j, int l) -> double { return (l == 0 && j == a.unit_row) ? 1.0 : 0.0; };
    double res0[NL], resN[NL];
    for (int l = 0; l < NL; ++l) {
        const double given = a.bv[l];
        const double fbN = (l == 0 && n - 1 == a.unit_row) ? 1.0 : 0.0, fb0 = (l == 0 && 0 == a.unit_row) ? 1.0 : 0.0;
        if (BC == 1) { res0[l] = given; resN[l] = fbN; }
        else { resN[l] = given; res0[l] = fb0; }
    }
    const int nmax = n - 2;
#ifndef V_NO_UNROLL
#pragma unroll
#endif
    for (int l = 0; l < NL; ++l) {
        auto rhs_row = [&](int j) -> double {
            if (j == 1) return res0[l] * RB(2, 2) + fv(1, l) * RB(2, 3) + fv(2, l) * RB(2, 4) + fv(3, l) * RB(2, 5);
            if (j == 2) return res0[l] * RB(3, 1) + fv(1, l) * RB(3, 2) + fv(2, l) * RB(3, 3) + fv(3, l) * RB(3, 4) + fv(4, l) * RB(3, 5);
            if (j == 3) return res0[l] * RB(4, 0) + fv(1, l) * RB(4, 1) + fv(2, l) * RB(4, 2) + fv(3, l) * RB(4, 3) + fv(4, l) * RB(4, 4) + fv(5, l) * RB(4, 5);
            if (j == n - 4)
                return fv(n - 6, l) * RT(0, 1) + fv(n - 5, l) * RT(0, 2) + fv(n - 4, l) * RT(0, 3) + fv(n - 3, l) * RT(0, 4) + fv(n - 2, l) * RT(0, 5) +
                       resN[l] * RT(0, 6);
            if (j == n - 3) return fv(n - 5, l) * RT(1, 1) + fv(n - 4, l) * RT(1, 2) + fv(n - 3, l) * RT(1, 3) + fv(n - 2, l) * RT(1, 4) + resN[l] * RT(1, 5);
            if (j == n - 2) return fv(n - 4, l) * RT(2, 1) + fv(n - 3, l) * RT(2, 2) + fv(n - 2, l) * RT(2, 3) + resN[l] * RT(2, 4);
            return fv(j - 2, l) * Rr(j, 1) + fv(j - 1, l) * Rr(j, 2) + fv(j, l) * Rr(j, 3) + fv(j + 1, l) + fv(j + 2, l) * Rr(j, 5);
        };
        const double bcs_b = res0[l] * RB(1, 3) + fv(1, l) * RB(1, 4) + fv(2, l) * RB(1, 5) + fv(3, l) * RB(1, 1);
        const double bcs_t = fv(n - 4, l) * RT(3, 5) + fv(n - 3, l) * RT(3, 1) + fv(n - 2, l) * RT(3, 2) + resN[l] * RT(3, 3);
#ifdef V_VOLATILE
        volatile double *y = a.scratch + (long long)l * n * nm + t;
        volatile double *x = a.dst + (long long)l * n * nm + t;
#else
        double *y = a.scratch + (long long)l * n * nm + t;
        double *x = a.dst + (long long)l * n * nm + t;
#endif
        auto Y = [&](int j) { return y[(long long)j * nm]; };
#ifdef V_NOUNROLL_M
#pragma clang loop unroll(disable)
#endif
        for (int m = 0; m < nmax; ++m) {
            const int j = m + 1;
            const double r = rhs_row(j);
            double v;
            if (m == 0) v = r * F(2, j);
            else if (m == 1) v = r - Y(j - 1) * F(2, j);
            else if (m == 2) v = r - Y(j - 1) * F(2, j) - Y(j - 2) * F(1, j);
            else v = r - Y(j - 1) * F(2, j) - Y(j - 2) * F(1, j) - Y(j - 3) * F(0, j);
            y[(long long)j * nm] = v;
        }
        auto XX = [&](int j) { return x[(long long)j * nm]; };
#ifdef V_NOUNROLL_M
#pragma clang loop unroll(disable)
#endif
        for (int m = nmax - 1; m >= 0; --m) {
            const int j = m + 1;
            const double yv = Y(j);
            double v;
            if (m == nmax - 1) v = yv / F(3, j);
            else if (m == nmax - 2) v = (yv - XX(j + 1) * F(4, j)) / F(3, j);
            else if (m == nmax - 3) v = (yv - XX(j + 1) * F(4, j) - XX(j + 2) * F(5, j)) / F(3, j);
            else v = (yv - XX(j + 1) * F(4, j) - XX(j + 2) * F(5, j) - XX(j + 3) * F(6, j)) / F(3, j);
            x[(long long)j * nm] = v;
        }
        auto X = [&](int j) { return x[(long long)j * nm]; };
        if (BC == 2) {
            double r0 = bcs_b;
            for (int ic = 1; ic <= idl - 1; ++ic) r0 = r0 + F(idl + ic - 1, 0) * X(ic);
            r0 = r0 + F(0, 0) * X(idl);
            x[0] = r0;
            x[(long long)(n - 1) * nm] = resN[l];
            double du = F(idl - 1, n - 1) * resN[l];
            for (int ic = 1; ic <= idl - 1; ++ic) du = du + F(idl - ic - 1, n - 1) * X(n - 1 - ic);
            du = du + F(ndi - 1, n - 1) * X(n - 1 - idl);
            for (int ic = 1; ic <= idr - 1; ++ic) du = du + Rr(n - 1, idr - ic) * fv(n - 1 - ic, l);
            a.du[(long long)l * nm + t] = du;
        } else {
            double rN = bcs_t;
            for (int ic = 1; ic <= idl - 1; ++ic) rN = rN + F(idl - ic - 1, n - 1) * X(n - 1 - ic);
            rN = rN + F(ndi - 1, n - 1) * X(n - 1 - idl);
            x[(long long)(n - 1) * nm] = rN;
            x[0] = res0[l];
            double du = F(idl - 1, 0) * res0[l];
            for (int ic = 1; ic <= idl - 1; ++ic) du = du + F(idl + ic - 1, 0) * X(ic);
            du = du + F(0, 0) * X(idl);
            for (int ic = 1; ic <= idr - 1; ++ic) du = du + Rr(0, idr + ic) * fv(ic, l);
            a.du[(long long)l * nm + t] = du;
        }
    }
}

#ifndef HOST_ONLY
template <int BC, int NL>
__global__ void __launch_bounds__(256) k(Args a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nm) return;
    body<BC, NL>(a, t);
}

#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static double rnd(unsigned long long &s) {      // splitmix64 -> (0, 1)
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((z >> 11) + 0.5) / 9007199254740992.0;
}

template <int BC>
static int run_case() {
    const int n = 40, NL = 2;
    const long long nm = 305;
    unsigned long long seed = 20250509 + BC;
    std::vector<double> fac((size_t)7 * n * nm), rb((size_t)40 * nm), rt((size_t)40 * nm), R((size_t)n * 5);
    for (auto &v : fac) v = 0.05 + 0.25 * rnd(seed);          // small off-diagonal factors: the recurrences stay bounded
    for (long long j = 0; j < n; ++j)
        for (long long t = 0; t < nm; ++t) fac[((size_t)3 * n + j) * nm + t] = 1.0 + rnd(seed);      // the pivot the backward sweep divides by
    for (auto &v : rb) v = rnd(seed) - 0.5;
    for (auto &v : rt) v = rnd(seed) - 0.5;
    for (auto &v : R) v = rnd(seed) - 0.5;
    std::vector<double> hs((size_t)5 * n * nm, 0.0), hd((size_t)NL * n * nm, 0.0), hu((size_t)NL * nm, 0.0);
    Args h{};
    h.n = n; h.nm = nm; h.unit_row = BC == 1 ? n - 1 : 0; h.bv[0] = 0.0; h.bv[1] = 1.0; h.bv[2] = 0.0;
    h.scratch = hs.data(); h.dst = hd.data(); h.du = hu.data(); h.g_fac = fac.data(); h.g_rb = rb.data(); h.g_rt = rt.data(); h.g_R = R.data();
    for (long long t = 0; t < nm; ++t) body<BC, NL>(h, t);
#ifdef HOST_ONLY
    double cs = 0.0;
    for (double v : hd) cs += v;
    printf("BC = %d: host replay finished, checksum %.17g\n", BC, cs);
    return 0;
#else
    Args d = h;
    double *dfac, *drb, *drt, *dR, *ds, *dd, *du;
    CK(hipMalloc(&dfac, fac.size() * 8)); CK(hipMalloc(&drb, rb.size() * 8)); CK(hipMalloc(&drt, rt.size() * 8)); CK(hipMalloc(&dR, R.size() * 8));
    CK(hipMalloc(&ds, hs.size() * 8)); CK(hipMalloc(&dd, hd.size() * 8)); CK(hipMalloc(&du, hu.size() * 8));
    CK(hipMemcpy(dfac, fac.data(), fac.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drb, rb.data(), rb.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(drt, rt.data(), rt.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(ds, 0, hs.size() * 8)); CK(hipMemset(dd, 0, hd.size() * 8)); CK(hipMemset(du, 0, hu.size() * 8));
    d.scratch = ds; d.dst = dd; d.du = du; d.g_fac = dfac; d.g_rb = drb; d.g_rt = drt; d.g_R = dR;
    hipLaunchKernelGGL((k<BC, NL>), dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, 0, d);
    CK(hipDeviceSynchronize());
    std::vector<double> gd(hd.size()), gu(hu.size());
    CK(hipMemcpy(gd.data(), dd, gd.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(gu.data(), du, gu.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    long long nbad = 0;
    for (size_t i = 0; i < gd.size(); ++i) {
        scale = std::fmax(scale, std::fabs(hd[i]));
        if (std::memcmp(&gd[i], &hd[i], 8) != 0) { ++nbad; worst = std::fmax(worst, std::fabs(gd[i] - hd[i])); }
    }
    for (size_t i = 0; i < gu.size(); ++i)
        if (std::memcmp(&gu[i], &hu[i], 8) != 0) { ++nbad; worst = std::fmax(worst, std::fabs(gu[i] - hu[i])); }
    printf("BC = %d: %lld of %zu values differ from the host, largest difference %.3e (largest value %.3e)\n", BC, nbad, gd.size() + gu.size(), worst, scale);
    return nbad != 0;
#endif
}

int main() {
    int bad = run_case<1>();
    bad += run_case<2>();
    printf(bad ? "MISMATCH\n" : "device == host, bit for bit\n");
    return bad;
}
