// OPR_FILTER_1D (operators/opr_filter.f90:393-460) on the device for the filter types COMPACT, 6E, 4E and COMPACT_CUTOFF -- the 1-D filters the
// dealiasing branch of OPR_Burgers_1D applies to the velocity and to ds/dx (physics/opr_burgers.f90:478-500).
//
// One thread per line, the reference's operations in the reference's order (src/filters/flt_compact.f90, flt_explitic.f90; TRIDSS / TRIDPSS
// utils/linear3.f90:56-150, 321-442; PENTADSS2 / PENTADPSS utils/linear5.f90:207-411), FP contraction off: correctness first -- no example of the
// reference selects [Dealiasing], and a fast form would be the chunked kernels with these right-hand sides.  The coefficient table f%coeffs is the
// host's (OPR_FILTER_INITIALIZE, opr_filter.f90:236-275; tlab_filter_create).  The tophat family (flt_tophat.f90) is not built: TLAB_EUNSUPPORTED.
#include <hip/hip_runtime.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tlab_amd.h"
#include "kernels.hpp"
#include "plan.hpp"
#include "profile.hpp"

extern "C" void tlab_internal_dealiasing_forget(tlab_filter_t f);      // capi.cpp
extern hipStream_t tlab_current_stream();
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();

using namespace tlab;

struct tlab_filter {
    int type, n, periodic, bcsmin, bcsmax, ncols;
    DeviceArray coeffs;      // [ncols][n], column-major like f%coeffs
    DeviceArray ws;          // two transposed copies of a field: x lines are filtered with the lines made the fastest index (tlab_internal_filter_1d)
};

namespace {

constexpr int FLT_COMPACT = 1, FLT_6E = 2, FLT_4E = 3, FLT_TOPHAT = 8, FLT_CUTOFF = 9;       // opr_filter.f90:56-65
constexpr int FBCS_ZERO = 6;                                                                   // filters/flt_base.f90:11

struct FilterArgs {
    const double *in;
    double *out;
    LineGeom g;
    int type, periodic, bcsmin, bcsmax;
    const double *c;         // device coeffs
};

__global__ void __launch_bounds__(256) k_filter1d(FilterArgs a) {
#pragma clang fp contract(off)
    const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= a.g.nlines) return;
    const int n = a.g.n;
    const long long rs = a.g.row_stride;
    const long long base = (line / a.g.lines_inner) * a.g.outer_stride + (line % a.g.lines_inner);
    const double *u = a.in + base;
    double *f = a.out + base;
#define U(i) u[(long long)((i)-1) * rs]
#define F(i) f[(long long)((i)-1) * rs]
#define C(i, k) a.c[((i)-1) + (size_t)n * ((k)-1)]
    const bool per = a.periodic != 0;
    if (a.type == FLT_COMPACT) {
        // ---- FLT_C4_RHS (flt_compact.f90:225-296) ----
        if (per) {
            F(1) = C(1, 1) * U(n - 1) + C(1, 2) * U(n) + C(1, 3) * U(1) + C(1, 4) * U(2) + C(1, 5) * U(3);
            F(2) = C(2, 1) * U(n) + C(2, 2) * U(1) + C(2, 3) * U(2) + C(2, 4) * U(3) + C(2, 5) * U(4);
            F(n) = C(n, 1) * U(n - 2) + C(n, 2) * U(n - 1) + C(n, 3) * U(n) + C(n, 4) * U(1) + C(n, 5) * U(2);
            F(n - 1) = C(n - 1, 1) * U(n - 3) + C(n - 1, 2) * U(n - 2) + C(n - 1, 3) * U(n - 1) + C(n - 1, 4) * U(n) + C(n - 1, 5) * U(1);
        } else {
            F(1) = C(1, 1) * U(1) + C(1, 2) * U(2) + C(1, 3) * U(3) + C(1, 4) * U(4) + C(1, 5) * U(5);
            F(2) = C(2, 1) * U(1) + C(2, 2) * U(2) + C(2, 3) * U(3) + C(2, 4) * U(4) + C(2, 5) * U(5);
            F(n - 1) = C(n - 1, 5) * U(n) + C(n - 1, 4) * U(n - 1) + C(n - 1, 3) * U(n - 2) + C(n - 1, 2) * U(n - 3) + C(n - 1, 1) * U(n - 4);
            F(n) = C(n, 5) * U(n) + C(n, 4) * U(n - 1) + C(n, 3) * U(n - 2) + C(n, 2) * U(n - 3) + C(n, 1) * U(n - 4);
            if (a.bcsmin == FBCS_ZERO) F(1) = U(1);
            if (a.bcsmax == FBCS_ZERO) F(n) = U(n);
        }
        for (int i = 3; i <= n - 2; ++i) F(i) = C(i, 1) * U(i - 2) + C(i, 2) * U(i - 1) + C(i, 3) * U(i) + C(i, 4) * U(i + 1) + C(i, 5) * U(i + 2);
        if (per) {      // TRIDPSS with (a, b, c, d, e) = coeffs(:, 6:10)   linear3.f90:321-442
            F(1) = F(1) * C(1, 7);
            for (int i = 2; i <= n - 1; ++i) F(i) = F(i) * C(i, 7) + C(i, 6) * F(i - 1);
            double wrk = 0.0;
            for (int i = 1; i <= n - 1; ++i) wrk = wrk + C(i, 9) * F(i);
            F(n) = (F(n) - wrk) * C(n, 7);
            F(n - 1) = C(n - 1, 10) * F(n) + F(n - 1);
            const double fn = F(n);
            for (int i = n - 2; i >= 1; --i) F(i) = F(i) + C(i, 8) * F(i + 1) + C(i, 10) * fn;
        } else {        // TRIDSS with (a, b, c) = coeffs(:, 6:8)   linear3.f90:56-150
            for (int i = 2; i <= n; ++i) F(i) = F(i) + C(i, 6) * F(i - 1);
            F(n) = F(n) * C(n, 7);
            for (int i = n - 1; i >= 1; --i) F(i) = (F(i) + C(i, 8) * F(i + 1)) * C(i, 7);
        }
    } else if (a.type == FLT_CUTOFF) {
        const double BD2 = 0.66059, CD2 = 0.1666774, DD2 = 0.679925e-3, CA = 0.9891856;      // flt_compact.f90:11-16
        if (per) {      // FLT_C4P_CUTOFF_RHS :327-349
            for (int i = 1; i <= n; ++i) {
                auto w = [&](int k) { int j = i + k; if (j < 1) j += n; if (j > n) j -= n; return U(j); };
                F(i) = BD2 * (w(1) + w(-1)) + CD2 * (w(2) + w(-2)) + DD2 * (w(3) + w(-3)) + CA * U(i);
            }
        } else {        // FLT_C4_CUTOFF_RHS :351-375
            F(1) = (15.0 * U(1) + 4.0 * U(2) - 6.0 * U(3) + 4.0 * U(4) - U(5)) / 16.0;
            F(2) = (12.0 * U(2) + U(1) + 6.0 * U(3) - 4.0 * U(4) + U(5)) / 16.0;
            F(3) = (10.0 * U(3) - U(1) + 4.0 * U(2) + 4.0 * U(4) - U(5)) / 16.0;
            F(n - 2) = (10.0 * U(n - 2) - U(n) + 4.0 * U(n - 1) + 4.0 * U(n - 3) - U(n - 4)) / 16.0;
            F(n - 1) = (12.0 * U(n - 1) + U(n) + 6.0 * U(n - 2) - 4.0 * U(n - 3) + U(n - 4)) / 16.0;
            F(n) = (15.0 * U(n) + 4.0 * U(n - 1) - 6.0 * U(n - 2) + 4.0 * U(n - 3) - U(n - 4)) / 16.0;
            for (int i = 4; i <= n - 3; ++i) F(i) = BD2 * (U(i + 1) + U(i - 1)) + CD2 * (U(i + 2) + U(i - 2)) + DD2 * (U(i + 3) + U(i - 3)) + CA * U(i);
        }
        // PENTADSS2 (linear5.f90:207-268) with (a .. e) = coeffs(:, 1:5); periodic: PENTADPSS (:352-411) with f, g = coeffs(:, 6:7)
        const double *A = a.c, *B = a.c + (size_t)n, *Cc = a.c + (size_t)2 * n, *D = a.c + (size_t)3 * n, *E = a.c + (size_t)4 * n;
        F(n - 1) = F(n - 1) - F(n) * D[n - 2];
        for (int i = n - 2; i >= 1; --i) F(i) = F(i) - F(i + 1) * D[i - 1] - F(i + 2) * E[i - 1];
        F(1) = F(1) / Cc[0];
        F(2) = (F(2) - F(1) * B[1]) / Cc[1];
        for (int i = 3; i <= n; ++i) F(i) = (F(i) - F(i - 1) * B[i - 1] - F(i - 2) * A[i - 1]) / Cc[i - 1];
        if (per) {
            const double *Fv = a.c + (size_t)5 * n, *Gv = a.c + (size_t)6 * n;
            const double m1 = E[n - 1] * Fv[0] + A[0] * Fv[n - 2] + B[0] * Fv[n - 1] + 1.0;
            const double m2 = E[n - 1] * Gv[0] + A[0] * Gv[n - 2] + B[0] * Gv[n - 1];
            const double m3 = D[n - 1] * Fv[0] + E[n - 1] * Fv[1] + A[0] * Fv[n - 1];
            const double m4 = D[n - 1] * Gv[0] + E[n - 1] * Gv[1] + A[0] * Gv[n - 1] + 1.0;
            const double di = 1 / (m1 * m4 - m2 * m3);
            const double d11 = di * (m4 * E[n - 1] - m2 * D[n - 1]), d12 = di * (m4 * B[0] - m2 * A[0]), d13 = di * m4 * A[0], d14 = di * m2 * E[n - 1];
            const double d21 = di * (m1 * D[n - 1] - m3 * E[n - 1]), d22 = di * (m1 * A[0] - m3 * B[0]), d23 = di * m3 * A[0], d24 = di * m1 * E[n - 1];
            const double dummy1 = d11 * F(1) + d12 * F(n) + d13 * F(n - 1) - d14 * F(2);
            const double dummy2 = d21 * F(1) + d22 * F(n) - d23 * F(n - 1) + d24 * F(2);
            for (int i = 3; i <= n - 3; ++i) F(i) = F(i) - dummy1 * Fv[i - 1] - dummy2 * Gv[i - 1];
            const int idx[5] = {1, 2, n - 2, n - 1, n};
            for (int q = 0; q < 5; ++q) F(idx[q]) = F(idx[q]) - dummy1 * Fv[idx[q] - 1] - dummy2 * Gv[idx[q] - 1];
        }
    } else if (a.type == FLT_6E) {      // FLT_E6 flt_explitic.f90:179-362
        const double b0 = 11.0 / 16.0, b1 = 15.0 / 64.0, b2 = -3.0 / 32.0, b3 = 1.0 / 64.0;
        const double bb[7] = {1.0 / 16.0, 3.0 / 4.0, 3.0 / 8.0, -1.0 / 4.0, 1.0 / 16.0, 0.0, 0.0};
        const double bc[7] = {-1.0 / 32.0, 5.0 / 32.0, 11.0 / 16.0, 5.0 / 16.0, -5.0 / 32.0, 1.0 / 32.0, 0.0};
        int ks = 1, ke = n;
        if (!per) {
            F(1) = U(1);
            if (a.bcsmin == 1) {
                F(2) = bb[0] * U(1) + bb[1] * U(2) + bb[2] * U(3) + bb[3] * U(4) + bb[4] * U(5) + bb[5] * U(6) + bb[6] * U(7);
                F(3) = bc[0] * U(1) + bc[1] * U(2) + bc[2] * U(3) + bc[3] * U(4) + bc[4] * U(5) + bc[5] * U(6) + bc[6] * U(7);
            } else {
                F(2) = U(2); F(3) = U(3);
            }
            ks = 4;
            F(n) = U(n);
            if (a.bcsmax == 1) {
                F(n - 1) = bb[0] * U(n) + bb[1] * U(n - 1) + bb[2] * U(n - 2) + bb[3] * U(n - 3) + bb[4] * U(n - 4) + bb[5] * U(n - 5) + bb[6] * U(n - 6);
                F(n - 2) = bc[0] * U(n) + bc[1] * U(n - 1) + bc[2] * U(n - 2) + bc[3] * U(n - 3) + bc[4] * U(n - 4) + bc[5] * U(n - 5) + bc[6] * U(n - 6);
            } else {
                F(n - 2) = U(n - 2); F(n - 1) = U(n - 1);
            }
            ke = n - 3;
        }
        for (int k = ks; k <= ke; ++k) {
            auto w = [&](int d) { int j = k + d; if (j < 1) j += n; if (j > n) j -= n; return U(j); };
            F(k) = b3 * (w(-3) + w(3)) + b2 * (w(-2) + w(2)) + b1 * (w(-1) + w(1)) + b0 * U(k);
        }
    } else {                            // FLT_E4 flt_explitic.f90:17-62
        if (per) {
            F(1) = C(1, 1) * U(n - 1) + C(1, 2) * U(n) + C(1, 3) * U(1) + C(1, 4) * U(2) + C(1, 5) * U(3);
            F(2) = C(2, 1) * U(n) + C(2, 2) * U(1) + C(2, 3) * U(2) + C(2, 4) * U(3) + C(2, 5) * U(4);
            F(n - 1) = C(n - 1, 1) * U(n - 3) + C(n - 1, 2) * U(n - 2) + C(n - 1, 3) * U(n - 1) + C(n - 1, 4) * U(n) + C(n - 1, 5) * U(1);
            F(n) = C(n, 1) * U(n - 2) + C(n, 2) * U(n - 1) + C(n, 3) * U(n) + C(n, 4) * U(1) + C(n, 5) * U(2);
        } else {
            F(1) = U(1);
            F(2) = C(2, 2) * U(1) + C(2, 3) * U(2) + C(2, 4) * U(3) + C(2, 5) * U(4) + C(2, 1) * U(5);
            F(n - 1) = C(n - 1, 1) * U(n - 3) + C(n - 1, 2) * U(n - 2) + C(n - 1, 3) * U(n - 1) + C(n - 1, 4) * U(n) + C(n - 1, 5) * U(n - 4);
            F(n) = U(n);
        }
        for (int i = 3; i <= n - 2; ++i) F(i) = C(i, 1) * U(i - 2) + C(i, 2) * U(i - 1) + C(i, 3) * U(i) + C(i, 4) * U(i + 1) + C(i, 5) * U(i + 2);
    }
#undef U
#undef F
#undef C
}

LineGeom geom_of(int dir, int nx, int ny, int nz) {
    LineGeom g;
    if (dir == 1) { g.n = nx; g.nlines = (long long)ny * nz; g.row_stride = 1; g.lines_inner = 1; g.outer_stride = nx; }
    else if (dir == 2) { g.n = ny; g.nlines = (long long)nx * nz; g.row_stride = nx; g.lines_inner = nx; g.outer_stride = (long long)nx * ny; }
    else { g.n = nz; g.nlines = (long long)nx * ny; g.row_stride = (long long)nx * ny; g.lines_inner = nx * ny; g.outer_stride = 0; }
    return g;
}

template <class F>
int guarded(F &&f) {
    try {
        f();
        return TLAB_OK;
    } catch (const std::invalid_argument &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    } catch (const std::domain_error &e) {
        tlab_set_error(e.what());
        return TLAB_EUNSUPPORTED;
    } catch (const std::bad_alloc &) {
        tlab_set_error("out of memory");
        return TLAB_ENOMEM;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EHIP;
    }
}

}  // namespace

void tlab_internal_filter_1d(int dir, tlab_filter_t f, int nx, int ny, int nz, const double *u, double *result, hipStream_t st) {
    const LineGeom g = geom_of(dir, nx, ny, nz);
    if (g.n != f->n) throw std::invalid_argument("filter size does not match the field size along dir");
    FilterArgs a;
    a.in = u; a.out = result; a.g = g; a.type = f->type; a.periodic = f->periodic; a.bcsmin = f->bcsmin; a.bcsmax = f->bcsmax; a.c = f->coeffs.p;
    if (dir == 1 && g.nlines >= 64) {
        // x lines: one thread per line would stride through contiguous memory (64 cache lines per wave access).  Like the reference (OPR_FILTER_X:
        // TLab_Transpose, filter, transpose back) the lines are made the fastest index first: two transposes at the copy rate + the coalesced filter
        const size_t N = (size_t)g.n * g.nlines;
        if (f->ws.n < 2 * N) f->ws.alloc(2 * N);
        double *t1 = f->ws.p, *t2 = t1 + N;
        if (launch_transpose(u, t1, g.n, (int)g.nlines, st) != hipSuccess) throw std::runtime_error("transpose");
        a.in = t1; a.out = t2;
        a.g.row_stride = g.nlines; a.g.lines_inner = (int)g.nlines; a.g.outer_stride = 0;
        {
            ProfScope ps("k_filter1d", st, (double)g.nlines * g.n * 16.0);
            hipLaunchKernelGGL(k_filter1d, dim3((unsigned)((g.nlines + 255) / 256)), dim3(256), 0, st, a);
        }
        if (hipGetLastError() != hipSuccess) throw std::runtime_error("k_filter1d launch failed");
        if (launch_transpose(t2, result, (int)g.nlines, g.n, st) != hipSuccess) throw std::runtime_error("transpose");
        return;
    }
    ProfScope ps("k_filter1d", st, (double)g.nlines * g.n * 16.0);
    hipLaunchKernelGGL(k_filter1d, dim3((unsigned)((g.nlines + 255) / 256)), dim3(256), 0, st, a);
    if (hipGetLastError() != hipSuccess) throw std::runtime_error("k_filter1d launch failed");
}

extern "C" {

int tlab_filter_create(tlab_filter_t *out, int type, int size, int periodic, int bcsmin, int bcsmax, int inb_filter, const double *coeffs) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("tlab_filter_create: null handle");
        if (type == FLT_TOPHAT) throw std::domain_error("tophat filters (flt_tophat.f90) are not built on the device path");
        if (type != FLT_COMPACT && type != FLT_6E && type != FLT_4E && type != FLT_CUTOFF)
            throw std::domain_error("filter type: compact (1), explicit6 (2), explicit4 (3), compactcutoff (9) -- the spectral / Helmholtz types are 3-D filters, not OPR_FILTER_1D");
        const int need = type == FLT_COMPACT ? 10 : type == FLT_4E ? 5 : type == FLT_CUTOFF ? (periodic ? 7 : 5) : 0;
        if (size < 8) throw std::invalid_argument("tlab_filter_create: at least 8 points");
        if (need > 0 && (!coeffs || inb_filter < need)) throw std::invalid_argument("tlab_filter_create: f%coeffs with at least inb_filter columns (opr_filter.f90:121-139)");
        if (!tlab_device_ready()) throw std::runtime_error("tlab_init has not been called (no CPU fallback exists)");
        auto f = std::make_unique<tlab_filter>();
        f->type = type; f->n = size; f->periodic = periodic ? 1 : 0; f->bcsmin = bcsmin; f->bcsmax = bcsmax; f->ncols = need;
        if (need > 0) f->coeffs.upload(std::vector<double>(coeffs, coeffs + (size_t)size * need));
        *out = f.release();
    });
}

int tlab_filter_destroy(tlab_filter_t f) {
    if (f) tlab_internal_dealiasing_forget(f);      // (a [PressureFilter] of a live tlab_dns must be taken out by its owner first: tlab_dns_set_pressure_filter)
    delete f;
    return TLAB_OK;
}

int tlab_opr_filter_1d(int dir, tlab_filter_t f, int nx, int ny, int nz, const double *u, double *result) {
    return guarded([&] {
        if (!f || !u || !result || u == result) throw std::invalid_argument("tlab_opr_filter_1d: bad arguments (out of place)");
        if (dir < 1 || dir > 3 || nx < 1 || ny < 1 || nz < 1) throw std::invalid_argument("tlab_opr_filter_1d: bad sizes");
        tlab_internal_filter_1d(dir, f, nx, ny, nz, u, result, tlab_current_stream());
    });
}

// OPR_FILTER (operators/opr_filter.f90:283-392), directional branch: x, y, z in that order, each `repeat` times, in place through tmp
int tlab_opr_filter(int nx, int ny, int nz, tlab_filter_t fx, tlab_filter_t fy, tlab_filter_t fz, const int *repeat, double *u, double *tmp) {
    return guarded([&] {
        if (!u || !tmp || u == tmp || nx < 1 || ny < 1 || nz < 1) throw std::invalid_argument("tlab_opr_filter: bad arguments");
        tlab_filter_t f[3] = {fx, fy, fz};
        hipStream_t st = tlab_current_stream();
        const size_t bytes = (size_t)nx * ny * nz * sizeof(double);
        for (int d = 0; d < 3; ++d) {
            if (!f[d]) continue;
            const int rep = repeat ? repeat[d] : 1;
            if (rep < 1) throw std::invalid_argument("tlab_opr_filter: Filter.Repeat must be positive (opr_filter.f90:198-204)");
            for (int r = 0; r < rep; ++r) {
                tlab_internal_filter_1d(d + 1, f[d], nx, ny, nz, u, tmp, st);
                if (hipMemcpyAsync(u, tmp, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) throw std::runtime_error("hipMemcpyAsync");
            }
        }
    });
}

}  // extern "C"
