// C ABI of include/tlab_amd.h: runtime, plans, operator dispatch.
#include "../../include/tlab_amd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "kernels.hpp"
#include "plan.hpp"

using namespace tlab;

namespace {

thread_local std::string g_err;
hipStream_t g_stream = nullptr;
int g_device = -1;
int g_last_path = 0;
int g_force_path = 0;
DeviceArray *g_ws = nullptr;  // grow-only scratch field

struct HipError : std::runtime_error {
    explicit HipError(const std::string &s) : std::runtime_error(s) {}
};
struct Unsupported : std::runtime_error {
    explicit Unsupported(const std::string &s) : std::runtime_error(s) {}
};
struct Invalid : std::runtime_error {
    explicit Invalid(const std::string &s) : std::runtime_error(s) {}
};

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw HipError(std::string(what) + ": " + hipGetErrorString(e));
}

template <class F>
int guarded(F &&f) {
    try {
        f();
        return TLAB_OK;
    } catch (const HipError &e) {
        g_err = e.what();
        return TLAB_EHIP;
    } catch (const Unsupported &e) {
        g_err = e.what();
        return TLAB_EUNSUPPORTED;
    } catch (const std::bad_alloc &) {
        g_err = "out of memory";
        return TLAB_ENOMEM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return TLAB_EINVAL;
    }
}

// result of a nested C-ABI call inside a guarded body: pass the failure on with its class
void ok_or_throw(int rc) {
    if (rc == TLAB_OK) return;
    if (rc == TLAB_EHIP) throw HipError(g_err);
    if (rc == TLAB_EUNSUPPORTED) throw Unsupported(g_err);
    throw Invalid(g_err);
}

double *workspace(size_t n) {
    if (!g_ws) g_ws = new DeviceArray();
    if (g_ws->n < n) {
        if (g_ws->p) hip_check(hipFree(g_ws->p), "hipFree");
        g_ws->p = nullptr;
        hip_check(hipMalloc((void **)&g_ws->p, n * sizeof(double)), "hipMalloc(workspace)");
        g_ws->n = n;
    }
    return g_ws->p;
}

}  // namespace

void tlab_internal_filter_1d(int dir, tlab_filter_t f, int nx, int ny, int nz, const double *u, double *result, hipStream_t st);      // filter.hip

// hooks for the other translation units (poisson.hip, rhs.hip)
int tlab_internal_deferred_flush();      // deferred.cpp: a recorded Runge-Kutta tail runs before anything else is enqueued (no-op unless tlab_deferred_enable)
int tlab_internal_deferred_take_error();
hipStream_t tlab_current_stream() {
    (void)tlab_internal_deferred_flush();
    return g_stream;
}
void tlab_set_error(const std::string &s) { g_err = s; }
bool tlab_device_ready() { return g_device >= 0; }

// ------------------------------------------------------------------------------------------------
// DeviceArray
// ------------------------------------------------------------------------------------------------
namespace tlab {
DeviceArray::~DeviceArray() {
    if (p) (void)hipFree(p);
}
void DeviceArray::upload(const std::vector<double> &h) {
    if (p) hip_check(hipFree(p), "hipFree");
    p = nullptr;
    n = h.size();
    if (n == 0) return;
    hip_check(hipMalloc((void **)&p, n * sizeof(double)), "hipMalloc(table)");
    hip_check(hipMemcpy(p, h.data(), n * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(table)");
}
void DeviceArray::alloc(size_t count) {
    if (p) hip_check(hipFree(p), "hipFree");
    p = nullptr;
    n = count;
    if (n) hip_check(hipMalloc((void **)&p, n * sizeof(double)), "hipMalloc(scratch)");
}
}  // namespace tlab

// ------------------------------------------------------------------------------------------------
// plan: matrices, stencils, cached device systems
// ------------------------------------------------------------------------------------------------
TriDiag tlab_fdm_plan::tridiag(int which, int ibc) const {
    const DerTables &d = (which == 1) ? t.der1 : t.der2;
    const int n = d.n;
    if (d.ndl != 3) throw Unsupported("only tridiagonal LHS schemes are built (CompactJacobian6Penta is not)");
    std::vector<double> lhs(d.lhs.begin(), d.lhs.begin() + (size_t)3 * n);
    TriDiag T;
    T.n = n;
    T.periodic = d.periodic;
    if (which == 1 && !d.periodic && ibc != BCS_DD) {
        double rb[4 * 8] = {0}, rt[5 * 7] = {0};
        fdm_bcs_neumann(ibc, n, 3, lhs.data(), d.ndr, d.rhs.data(), rb, rt);
    }
    T.a.assign(lhs.begin(), lhs.begin() + n);
    T.b.assign(lhs.begin() + n, lhs.begin() + 2 * n);
    T.c.assign(lhs.begin() + 2 * n, lhs.begin() + 3 * n);
    if (which == 1 && !d.periodic) {
        if (ibc == BCS_ND || ibc == BCS_NN) {  // wall value forced to zero (fdm_derivative.f90:239-241): identity row
            T.a[0] = 0; T.b[0] = 1; T.c[0] = 0; T.a[1] = 0;
        }
        if (ibc == BCS_DN || ibc == BCS_NN) {
            T.a[n - 1] = 0; T.b[n - 1] = 1; T.c[n - 1] = 0; T.c[n - 2] = 0;
        }
    }
    return T;
}

StencilDev tlab_fdm_plan::stencil(int which, int ibc) {
    const DerTables &d = (which == 1) ? t.der1 : t.der2;
    const int nx = d.n;
    StencilDev s;
    std::memset(&s, 0, sizeof(s));
    s.sym = (which == 2);
    s.periodic = d.periodic ? 1 : 0;
    const double *r = d.rhs.data();
#define RI(i, k) r[((i)-1) + (size_t)nx * ((k)-1)]
    if (which == 1 && d.direct) {   // MatMul_3d / MatMul_5d (fdm_matmul.f90:70-121, 265-319) with the Neumann-reduced rows of the variant
        if ((d.ndr != 3 && d.ndr != 5) || d.periodic) throw Unsupported("direct first derivative: 3 or 5 RHS diagonals in a non-periodic direction");
        double rb[4 * 8] = {0}, rt[5 * 7] = {0};
        const bool nb = (ibc == BCS_ND || ibc == BCS_NN), ntp = (ibc == BCS_DN || ibc == BCS_NN);
        if (nb || ntp) {
            std::vector<double> lhs(d.lhs.begin(), d.lhs.begin() + (size_t)3 * nx);
            fdm_bcs_neumann(ibc, nx, 3, lhs.data(), d.ndr, d.rhs.data(), rb, rt);
        }
#define RB(j, c) rb[((j)-1) + 4 * (c)]
#define RT(rr, c) rt[(rr) + 5 * ((c)-1)]
        std::vector<double> rc((size_t)5 * nx, 0.0);
        if (d.ndr == 5) {
            for (int i = 1; i <= nx; ++i)
                for (int k = 1; k <= 5; ++k) rc[(size_t)(i - 1) * 5 + (k - 1)] = RI(i, k);
            for (int i = 5; i <= nx - 4; ++i) rc[(size_t)(i - 1) * 5 + 3] = 1.0;       // the interior loop has no r4 factor (:303-305)
            if (nb) {       // f(1) carries the boundary value, 0 (fdm_derivative.f90:240): its terms are dropped
                s.bb[1][1] = RB(2, 3); s.bb[1][2] = RB(2, 4); s.bb[1][3] = RB(2, 5);
                s.bb[2][1] = RB(3, 2); s.bb[2][2] = RB(3, 3); s.bb[2][3] = RB(3, 4); s.bb[2][4] = RB(3, 5);
                for (int k = 1; k <= 5; ++k) rc[(size_t)3 * 5 + (k - 1)] = RB(4, k);
            } else {
                s.bb[0][0] = RI(1, 3); s.bb[0][1] = RI(1, 4); s.bb[0][2] = RI(1, 5); s.bb[0][3] = RI(1, 1);
                s.bb[1][0] = RI(2, 2); s.bb[1][1] = RI(2, 3); s.bb[1][2] = RI(2, 4); s.bb[1][3] = RI(2, 5);
                for (int k = 0; k < 5; ++k) s.bb[2][k] = RI(3, 1 + k);
            }
            if (ntp) {
                for (int k = 1; k <= 5; ++k) rc[(size_t)(nx - 4) * 5 + (k - 1)] = RT(0, k);
                s.bt[0][1] = RT(1, 1); s.bt[0][2] = RT(1, 2); s.bt[0][3] = RT(1, 3); s.bt[0][4] = RT(1, 4);
                s.bt[1][2] = RT(2, 1); s.bt[1][3] = RT(2, 2); s.bt[1][4] = RT(2, 3);
            } else {
                for (int k = 0; k < 5; ++k) s.bt[0][1 + k] = RI(nx - 2, 1 + k);
                for (int k = 0; k < 4; ++k) s.bt[1][2 + k] = RI(nx - 1, 1 + k);
                s.bt[2][2] = RI(nx, 5); s.bt[2][3] = RI(nx, 1); s.bt[2][4] = RI(nx, 2); s.bt[2][5] = RI(nx, 3);
            }
        } else {
            for (int i = 1; i <= nx; ++i) {       // (0, r1, r2, 1, 0): the interior loop has no r3 factor (:103-105)
                rc[(size_t)(i - 1) * 5 + 1] = RI(i, 1); rc[(size_t)(i - 1) * 5 + 2] = RI(i, 2); rc[(size_t)(i - 1) * 5 + 3] = 1.0;
            }
            if (nb) {
                s.bb[1][1] = RB(2, 2); s.bb[1][2] = RB(2, 3);
                s.bb[2][1] = RB(3, 1); s.bb[2][2] = RB(3, 2); s.bb[2][3] = RB(3, 3);
            } else {
                s.bb[0][0] = RI(1, 2); s.bb[0][1] = RI(1, 3); s.bb[0][2] = RI(1, 1);
                s.bb[1][0] = RI(2, 1); s.bb[1][1] = RI(2, 2); s.bb[1][2] = RI(2, 3);
                s.bb[2][1] = RI(3, 1); s.bb[2][2] = RI(3, 2); s.bb[2][3] = RI(3, 3);
            }
            if (ntp) {
                s.bt[0][2] = RT(0, 1); s.bt[0][3] = RT(0, 2); s.bt[0][4] = RT(0, 3);
                s.bt[1][3] = RT(1, 1); s.bt[1][4] = RT(1, 2);
            } else {
                s.bt[0][2] = RI(nx - 2, 1); s.bt[0][3] = RI(nx - 2, 2); s.bt[0][4] = RI(nx - 2, 3);
                s.bt[1][3] = RI(nx - 1, 1); s.bt[1][4] = RI(nx - 1, 2); s.bt[1][5] = RI(nx - 1, 3);
                s.bt[2][3] = RI(nx, 3); s.bt[2][4] = RI(nx, 1); s.bt[2][5] = RI(nx, 2);
            }
        }
#undef RB
#undef RT
        auto &slot = rowc1[ibc & 3];
        if (!slot) {
            slot = std::make_unique<DeviceArray>();
            slot->upload(rc);
        }
        s.rowc = slot->p;
    } else if (which == 1) {
        if (d.ndr != 3 && d.ndr != 5) throw Unsupported("first-derivative RHS must have 3 or 5 diagonals");
        s.c2 = (d.ndr == 5) ? RI(4, 5) : 0.0;  // r5_loc, fdm_matmul.f90:373
        if (!d.periodic) {
            double rb[4 * 8] = {0}, rt[5 * 7] = {0};
            const bool nb = (ibc == BCS_ND || ibc == BCS_NN), ntp = (ibc == BCS_DN || ibc == BCS_NN);
            if (nb || ntp) {
                std::vector<double> lhs(d.lhs.begin(), d.lhs.begin() + (size_t)3 * nx);
                fdm_bcs_neumann(ibc, nx, 3, lhs.data(), d.ndr, d.rhs.data(), rb, rt);
            }
#define RB(j, c) rb[((j)-1) + 4 * (c)]
#define RT(rr, c) rt[(rr) + 5 * ((c)-1)]
            if (d.ndr == 5) {  // MatMul_5d_antisym :382-391, :406-415
                if (nb) {
                    s.bb[1][1] = RB(2, 3); s.bb[1][2] = RB(2, 4); s.bb[1][3] = RB(2, 5);
                    s.bb[2][1] = RB(3, 2); s.bb[2][2] = RB(3, 3); s.bb[2][3] = RB(3, 4); s.bb[2][4] = RB(3, 5);
                } else {
                    s.bb[0][0] = RI(1, 3); s.bb[0][1] = RI(1, 4); s.bb[0][2] = RI(1, 5); s.bb[0][3] = RI(1, 1);
                    s.bb[1][0] = RI(2, 2); s.bb[1][1] = RI(2, 3); s.bb[1][2] = RI(2, 4); s.bb[1][3] = RI(2, 5);
                    s.bb[2][0] = RI(3, 1); s.bb[2][1] = RI(3, 2); s.bb[2][2] = RI(3, 3); s.bb[2][3] = RI(3, 4); s.bb[2][4] = RI(3, 5);
                }
                if (ntp) {
                    s.bt[0][1] = RT(1, 1); s.bt[0][2] = RT(1, 2); s.bt[0][3] = RT(1, 3); s.bt[0][4] = RT(1, 4);
                    s.bt[1][2] = RT(2, 1); s.bt[1][3] = RT(2, 2); s.bt[1][4] = RT(2, 3);
                } else {
                    s.bt[0][1] = RI(nx - 2, 1); s.bt[0][2] = RI(nx - 2, 2); s.bt[0][3] = RI(nx - 2, 3); s.bt[0][4] = RI(nx - 2, 4); s.bt[0][5] = RI(nx - 2, 5);
                    s.bt[1][2] = RI(nx - 1, 1); s.bt[1][3] = RI(nx - 1, 2); s.bt[1][4] = RI(nx - 1, 3); s.bt[1][5] = RI(nx - 1, 4);
                    s.bt[2][2] = RI(nx, 5); s.bt[2][3] = RI(nx, 1); s.bt[2][4] = RI(nx, 2); s.bt[2][5] = RI(nx, 3);
                }
            } else {  // MatMul_3d_antisym :177-186, :200-209 ; third row from each wall is an interior row
                s.bb[2][1] = -1.0; s.bb[2][3] = 1.0;
                s.bt[0][2] = -1.0; s.bt[0][4] = 1.0;
                if (nb) {
                    s.bb[1][1] = RB(2, 2); s.bb[1][2] = RB(2, 3);
                } else {
                    s.bb[0][0] = RI(1, 2); s.bb[0][1] = RI(1, 3); s.bb[0][2] = RI(1, 1);
                    s.bb[1][0] = RI(2, 1); s.bb[1][1] = RI(2, 2); s.bb[1][2] = RI(2, 3);
                }
                if (ntp) {
                    s.bt[1][3] = RT(1, 1); s.bt[1][4] = RT(1, 2);
                } else {
                    s.bt[1][3] = RI(nx - 1, 1); s.bt[1][4] = RI(nx - 1, 2); s.bt[1][5] = RI(nx - 1, 3);
                    s.bt[2][3] = RI(nx, 3); s.bt[2][4] = RI(nx, 1); s.bt[2][5] = RI(nx, 2);
                }
            }
#undef RB
#undef RT
        }
    } else if (d.direct) {   // MatMul_5d, fdm_matmul.f90:291-318 (no boundary data: FDM_Der2_Solve passes BCS_DD)
        if (d.ndr != 5 || d.periodic) throw Unsupported("direct schemes: pentadiagonal RHS in a non-periodic direction only");
        s.bb[0][0] = RI(1, 3); s.bb[0][1] = RI(1, 4); s.bb[0][2] = RI(1, 5); s.bb[0][3] = RI(1, 1);
        s.bb[1][0] = RI(2, 2); s.bb[1][1] = RI(2, 3); s.bb[1][2] = RI(2, 4); s.bb[1][3] = RI(2, 5);
        for (int k = 0; k < 5; ++k) s.bb[2][k] = RI(3, 1 + k);
        for (int k = 0; k < 5; ++k) s.bt[0][1 + k] = RI(nx - 2, 1 + k);
        for (int k = 0; k < 4; ++k) s.bt[1][2 + k] = RI(nx - 1, 1 + k);
        s.bt[2][2] = RI(nx, 5); s.bt[2][3] = RI(nx, 1); s.bt[2][4] = RI(nx, 2); s.bt[2][5] = RI(nx, 3);
        if (!rowc2) {
            std::vector<double> rc((size_t)5 * nx, 0.0);
            for (int i = 1; i <= nx; ++i)
                for (int k = 1; k <= 5; ++k) rc[(size_t)(i - 1) * 5 + (k - 1)] = RI(i, k);
            for (int i = 5; i <= nx - 4; ++i) rc[(size_t)(i - 1) * 5 + 3] = 1.0;       // the interior loop has no r4 factor (:303-305)
            rowc2 = std::make_unique<DeviceArray>();
            rowc2->upload(rc);
        }
        s.rowc = rowc2->p;
    } else {
        if (d.ndr == 7) {  // MatMul_7d_sym :578-580, :596-601, :628-635
            s.c0 = RI(4, 4); s.c2 = RI(4, 6); s.c3 = RI(4, 7);
            if (!d.periodic) {
                s.bb[0][0] = RI(1, 4); s.bb[0][1] = RI(1, 5); s.bb[0][2] = RI(1, 6); s.bb[0][3] = RI(1, 7); s.bb[0][4] = RI(1, 1);
                s.bb[1][0] = RI(2, 3); s.bb[1][1] = RI(2, 4); s.bb[1][2] = RI(2, 5); s.bb[1][3] = RI(2, 6); s.bb[1][4] = RI(2, 7);
                for (int k = 0; k < 6; ++k) s.bb[2][k] = RI(3, 2 + k);
                for (int k = 0; k < 6; ++k) s.bt[0][k] = RI(nx - 2, 1 + k);
                for (int k = 0; k < 5; ++k) s.bt[1][1 + k] = RI(nx - 1, 1 + k);
                s.bt[2][1] = RI(nx, 7); s.bt[2][2] = RI(nx, 1); s.bt[2][3] = RI(nx, 2); s.bt[2][4] = RI(nx, 3); s.bt[2][5] = RI(nx, 4);
            }
        } else if (d.ndr == 5) {  // MatMul_5d_sym :438-439, :450-453, :474-478
            s.c0 = RI(3, 3); s.c2 = RI(3, 5); s.c3 = 0.0;
            if (!d.periodic) {
                s.bb[0][0] = RI(1, 3); s.bb[0][1] = RI(1, 4); s.bb[0][2] = RI(1, 5); s.bb[0][3] = RI(1, 1);
                s.bb[1][0] = RI(2, 2); s.bb[1][1] = RI(2, 3); s.bb[1][2] = RI(2, 4); s.bb[1][3] = RI(2, 5);
                s.bb[2][0] = s.c2; s.bb[2][1] = 1.0; s.bb[2][2] = s.c0; s.bb[2][3] = 1.0; s.bb[2][4] = s.c2;
                s.bt[0][1] = s.c2; s.bt[0][2] = 1.0; s.bt[0][3] = s.c0; s.bt[0][4] = 1.0; s.bt[0][5] = s.c2;
                s.bt[1][2] = RI(nx - 1, 1); s.bt[1][3] = RI(nx - 1, 2); s.bt[1][4] = RI(nx - 1, 3); s.bt[1][5] = RI(nx - 1, 4);
                s.bt[2][2] = RI(nx, 5); s.bt[2][3] = RI(nx, 1); s.bt[2][4] = RI(nx, 2); s.bt[2][5] = RI(nx, 3);
            }
        } else {
            throw Unsupported("second-derivative RHS must have 5 or 7 diagonals");
        }
    }
#undef RI
    return s;
}

SystemEntry &tlab_fdm_plan::system(int which, int ibc, int P) {
    if (which == 2 || t.der1.periodic) ibc = 0;
    auto key = std::make_tuple(which, ibc, P);
    auto it = systems.find(key);
    if (it != systems.end()) return *it->second;
    auto e = std::make_unique<SystemEntry>();
    TriDiag T = tridiag(which, ibc);
    build_chunked(T, P, e->host);
    const ChunkedTables &h = e->host;
    const int n = h.n, m = h.m;
    std::vector<double> rowtab((size_t)5 * n);
    std::copy(h.Lm.begin(), h.Lm.end(), rowtab.begin());
    std::copy(h.Dinv.begin(), h.Dinv.end(), rowtab.begin() + n);
    std::copy(h.Cm.begin(), h.Cm.end(), rowtab.begin() + 2 * n);
    std::copy(h.V.begin(), h.V.end(), rowtab.begin() + 3 * n);
    std::copy(h.W.begin(), h.W.end(), rowtab.begin() + 4 * n);
    e->rowtab.upload(rowtab);
    // lane-invariant: all chunks carry bitwise identical tables (circulant matrix with exactly uniform coefficients), so that the
    // kernel can use wave-uniform scalar loads.  NOTE: plans made by FDM_CreatePlan are NOT invariant: the reference derives the
    // "uniform" spacing numerically (fdm.f90:194-201) and its a,b,c wander by ~eps*n (2e-13 at n = 512).  Replacing them by chunk 0's
    // values would cost that much parity, so only exact equality qualifies; otherwise the per-lane tables are staged through LDS.
    auto close = [](double a, double b) { return a == b; };
    bool inv = h.periodic;
    for (int tab = 0; tab < 5 && inv; ++tab)
        for (int j = 1; j < P && inv; ++j)
            for (int p = 0; p < m; ++p)
                if (!close(rowtab[(size_t)tab * n + j * m + p], rowtab[(size_t)tab * n + p])) { inv = false; break; }
    if (P == 64) {      // wave-per-line kernel: parallel cyclic reduction over the 64 lanes
        if (h.pcr_steps != 6) throw std::runtime_error("internal: PCR schedule missing");
        std::vector<double> red((size_t)13 * 64);
        std::copy(h.pcr_k1.begin(), h.pcr_k1.end(), red.begin());
        std::copy(h.pcr_k2.begin(), h.pcr_k2.end(), red.begin() + 6 * 64);
        std::copy(h.pcr_dinv.begin(), h.pcr_dinv.end(), red.begin() + 12 * 64);
        for (int q = 0; q < 13 && inv; ++q)
            for (int j = 1; j < 64; ++j)
                if (!close(red[(size_t)q * 64 + j], red[(size_t)q * 64])) { inv = false; break; }
        e->red.upload(red);
    } else if (P == 128 || P == 256) {      // several waves per line: two-level reduction tables [21][P] (chunked.hpp), per-lane by construction
        if (h.tl_waves * 64 != P) throw std::runtime_error("two-level separator tables unavailable for this system");      // (inv: the row tables only)
        e->red.upload(h.tl);
    } else {
        e->red.upload(h.ginv);
    }
    e->lane_invariant = inv;
    {   // interior chunks equal to chunk 1 to the bit (the first and the last chunk carry the wall rows of a non-periodic system)
        bool ci = P >= 3;
        long long worst = 0;
        for (int tab = 0; tab < 5; ++tab)
            for (int j = 2; j < P - 1; ++j)
                for (int p = 0; p < m; ++p) {
                    const double u = rowtab[(size_t)tab * n + j * m + p], v = rowtab[(size_t)tab * n + m + p];
                    if (u != v) {
                        ci = false;
                        long long iu, iv;
                        std::memcpy(&iu, &u, 8);
                        std::memcpy(&iv, &v, 8);
                        worst = std::max(worst, std::llabs(iu - iv));
                    }
                }
        e->chunk_invariant = ci;
        bool band = P >= 8 && P <= 32 && (P & (P - 1)) == 0 && h.ginv.size() == (size_t)P * P;
        if (band) {
            double dmax = 0.0, omax = 0.0;
            std::vector<double> gb((size_t)P * 5, 0.0);
            for (int c = 0; c < P; ++c)
                for (int q = 0; q < P; ++q) {
                    int d = q - c;
                    if (d > P / 2) d -= P;
                    if (d < -P / 2) d += P;
                    const double v = h.ginv[(size_t)c * P + q];
                    if (d >= -2 && d <= 2) { gb[(size_t)c * 5 + d + 2] = v; if (d == 0) dmax = std::max(dmax, std::fabs(v)); }
                    else omax = std::max(omax, std::fabs(v));
                }
            band = omax <= 1e-30 * dmax;
            if (band) e->band.upload(gb);
        }
        e->band_ok = band;
        if (getenv("TLAB_DEBUG_TABLES")) fprintf(stderr, "system n=%d P=%d periodic=%d: lane_invariant %d chunk_invariant %d (largest interior difference %lld ulp) banded inverse %d\n", n, P, (int)h.periodic, (int)inv, (int)ci, worst, (int)band);
    }
    SystemEntry &ref = *e;
    systems[key] = std::move(e);
    return ref;
}

JacCorrDev tlab_fdm_plan::jaccorr() {
    if (!t.der2.need_1der) return JacCorrDev{nullptr};
    if (!jc) {
        jc = std::make_unique<DeviceArray>();
        const int n = t.n, ndr = t.der2.ndr;
        std::vector<double> j(t.der2.rhs.begin() + (size_t)n * ndr, t.der2.rhs.begin() + (size_t)n * (ndr + 3));
        jc->upload(j);
    }
    return JacCorrDev{jc->p};
}

// ------------------------------------------------------------------------------------------------
// runtime
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *tlab_last_error(void) { return g_err.c_str(); }

int tlab_init(int device) {
    return guarded([&] {
        int count = 0;
        hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
        if (count <= 0) throw HipError("no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)");
        if (device < 0 || device >= count) throw Invalid("tlab_init: bad device index");
        hip_check(hipSetDevice(device), "hipSetDevice");
        g_device = device;
    });
}

int tlab_finalize(void) {
    (void)tlab_deferred_enable(0);      // (runs what is still recorded)
    return guarded([&] {
        delete g_ws;
        g_ws = nullptr;
    });
}

int tlab_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int tlab_set_stream(void *s) {
    (void)tlab_internal_deferred_flush();
    g_stream = (hipStream_t)s;
    return TLAB_OK;
}

int tlab_sync(void) {
    const int rcd = tlab_internal_deferred_flush();
    const int rco = tlab_internal_deferred_take_error();      // a recorded substep that failed when another entry point made it run
    if (rcd != TLAB_OK) return rcd;
    if (rco != TLAB_OK) return rco;
    return guarded([&] { hip_check(hipStreamSynchronize(g_stream), "hipStreamSynchronize"); });
}

int tlab_malloc(void **p, size_t bytes) {
    return guarded([&] {
        const hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) (void)hipGetLastError();      // a refused allocation is an answer, not a fault: the next launch's error check must not find it
        hip_check(e, "hipMalloc");                          // (TLab_AMD_Place_Arrays asks for candidates until the memory says no)
    });
}
int tlab_free(void *p) {
    (void)tlab_internal_deferred_flush();
    return guarded([&] { hip_check(hipFree(p), "hipFree"); });
}
int tlab_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    const int rcd = tlab_internal_deferred_flush();
    if (rcd != TLAB_OK) return rcd;
    return guarded([&] {
        hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_stream), "hipMemcpy h2d");
        hip_check(hipStreamSynchronize(g_stream), "sync");
    });
}
int tlab_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    const int rcd = tlab_internal_deferred_flush();
    if (rcd != TLAB_OK) return rcd;
    return guarded([&] {
        hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_stream), "hipMemcpy d2h");
        hip_check(hipStreamSynchronize(g_stream), "sync");
    });
}

// ------------------------------------------------------------------------------------------------
// plans
// ------------------------------------------------------------------------------------------------
int tlab_fdm_plan_create(tlab_fdm_plan_t *out, int n, const double *nodes, int periodic, int uniform, int scheme1,
                         int scheme2, double hyper_bc1_ext) {
    return guarded([&] {
        if (!out || !nodes || n < 1) throw Invalid("tlab_fdm_plan_create: bad arguments");
        if (periodic && !uniform) throw Invalid("grid must be uniform in a periodic direction (fdm.f90:117-120)");
        if (n > 1 && n < 8) throw Invalid("tlab_fdm_plan_create: a direction needs 1 or >= 8 points");
        auto p = std::make_unique<tlab_fdm_plan>();
        try {
            fdm_create_plan(p->t, n, nodes, periodic != 0, uniform != 0, scheme1, scheme2, hyper_bc1_ext);
        } catch (const std::runtime_error &e) {
            throw Unsupported(e.what());
        }
        *out = p.release();
    });
}

int tlab_fdm_plan_create_from_arrays(tlab_fdm_plan_t *out, int n, int periodic, int need_1der, int ndl1, int ndr1,
                                     const double *lhs1, const double *rhs1, int ndl2, int ndr2, const double *lhs2,
                                     const double *rhs2) {
    return guarded([&] {
        if (!out || n < 8 || !lhs1 || !rhs1 || !lhs2 || !rhs2) throw Invalid("tlab_fdm_plan_create_from_arrays: bad arguments");
        const bool penta = (ndl1 == 5 && ndr1 == 7);      // CompactJacobian6Penta first derivative (generic kernel, reference operation order)
        if ((ndl1 != 3 && !penta) || ndl2 != 3) throw Unsupported("LHS: 3 diagonals (first derivative: or 5 with 7 RHS diagonals, CompactJacobian6Penta)");
        if ((!penta && ndr1 != 3 && ndr1 != 5) || (ndr2 != 5 && ndr2 != 7)) throw Unsupported("unsupported number of RHS diagonals");
        auto p = std::make_unique<tlab_fdm_plan>();
        FdmTables &t = p->t;
        t.n = n; t.periodic = periodic != 0; t.uniform = !need_1der;
        for (DerTables *d : {&t.der1, &t.der2}) { d->n = n; d->periodic = t.periodic; d->lhs.assign((size_t)n * 5, 0.0); d->mwn.assign(n, 0.0); }
        t.der1.ndl = ndl1; t.der1.ndr = ndr1; t.der1.rhs_cols = 7; t.der1.rhs.assign((size_t)n * 7, 0.0);
        t.der2.ndl = 3; t.der2.ndr = ndr2; t.der2.rhs_cols = 12; t.der2.rhs.assign((size_t)n * 12, 0.0);
        t.der2.need_1der = need_1der != 0;
        std::copy(lhs1, lhs1 + (size_t)n * ndl1, t.der1.lhs.begin());
        std::copy(rhs1, rhs1 + (size_t)n * ndr1, t.der1.rhs.begin());
        std::copy(lhs2, lhs2 + (size_t)n * 3, t.der2.lhs.begin());
        std::copy(rhs2, rhs2 + (size_t)n * (ndr2 + 3), t.der2.rhs.begin());
        if (penta) {      // the kernel follows the reference's PENTADSS2 / PENTADPSS: factorize the host's lhs the reference's way (fdm_derivative.f90:78-119)
            t.der1.mode_fdm = FDM_COM6_JACOBIAN_PENTA;
            const int all[4] = {BCS_DD, BCS_ND, BCS_DN, BCS_NN};
            try { der1_factorize(t.der1, all, 4); } catch (const std::runtime_error &e) { throw Unsupported(e.what()); }
        }
        *out = p.release();
    });
}

// what else the Poisson solver and the monitors take from type(fdm_dt): modified wavenumbers (fdm_derivative.f90:193-211, periodic
// directions; opr_elliptic.f90:199-203), Jacobian (fdm.f90:194-224; time.f90:148-152), node positions
int tlab_fdm_plan_set_aux(tlab_fdm_plan_t p, const double *mwn1, const double *mwn2, const double *jac, const double *nodes) {
    return guarded([&] {
        if (!p) throw Invalid("tlab_fdm_plan_set_aux: null plan");
        const int n = p->t.n;
        if (mwn1) p->t.der1.mwn.assign(mwn1, mwn1 + n);
        if (mwn2) p->t.der2.mwn.assign(mwn2, mwn2 + n);
        if (jac) p->t.jac.assign(jac, jac + (size_t)3 * n);
        if (nodes) p->t.nodes.assign(nodes, nodes + n);
    });
}

// mode_fdm of the two derivatives of a host-built plan (FDM_COM6_DIRECT = 16, FDM_COM4_DIRECT = 17 change how rhs is read)
int tlab_fdm_plan_set_scheme(tlab_fdm_plan_t p, int mode1, int mode2) {
    return guarded([&] {
        if (!p) throw Invalid("tlab_fdm_plan_set_scheme: null plan");
        auto direct = [](int m) { return m == FDM_COM6_DIRECT || m == FDM_COM4_DIRECT; };
        if (direct(mode1) && (p->t.periodic || p->t.der1.ndl != 3 || (p->t.der1.ndr != 3 && p->t.der1.ndr != 5)))
            throw Invalid("direct first derivative: non-periodic direction, tridiagonal LHS, 3 or 5 RHS diagonals (FDM_C1N4_Direct / FDM_C1N6_Direct)");
        if (direct(mode2) && (p->t.periodic || p->t.der2.ndr != 5)) throw Invalid("direct second derivative: non-periodic direction with 5 RHS diagonals");
        p->t.der1.mode_fdm = mode1; p->t.der2.mode_fdm = mode2;
        p->t.der1.direct = direct(mode1);
        p->t.der2.direct = direct(mode2);
        if (p->t.der2.direct) p->t.der2.need_1der = false;                 // fdm_derivative.f90:379,383
        p->systems.clear();
        p->rowc2.reset();
        for (auto &r : p->rowc1) r.reset();
    });
}

// [Staggering] StaggerHorizontalPressure (TLab_WorkFlow::stagger_on): what FDM_CreatePlan adds for a periodic direction (fdm.f90:236-248)
int tlab_fdm_plan_set_stagger(tlab_fdm_plan_t p, int mode) {
    return guarded([&] {
        if (!p) throw Invalid("tlab_fdm_plan_set_stagger: null plan");
        if (mode < 0 || mode > 2) throw Invalid("tlab_fdm_plan_set_stagger: mode 0 off, 1 tables + interpolatory wavenumbers, 2 tables only");
        for (auto &f : p->interp)
            if (f) { tlab_filter_destroy(f); f = nullptr; }
        if (mode == 0) { p->t.stagger = false; return; }
        if (p->t.jac.size() < (size_t)p->t.n) throw Invalid("tlab_fdm_plan_set_stagger: the plan has no Jacobian (tlab_fdm_plan_set_aux)");
        try { interpol_initialize(p->t, mode == 1); } catch (const std::runtime_error &e) { throw Unsupported(e.what()); }
    });
}

int tlab_fdm_plan_destroy(tlab_fdm_plan_t p) {
    if (p)
        for (auto &f : p->interp)
            if (f) tlab_filter_destroy(f);
    delete p;
    return TLAB_OK;
}

namespace { int xline_chunks(int n, tlab_fdm_plan_t g); }
int tlab_fdm_plan_info(tlab_fdm_plan_t p, int what) {
    if (!p) return TLAB_EINVAL;
    if (what == 8 || what == 9) {      // the x-line kernel's view of the plan (builds the chunked tables on first use: needs tlab_init)
        try {
            const int P = xline_chunks(p->t.n, p);
            if (what == 8 || P == 0) return P;
            return (p->system(1, 0, P).lane_invariant && p->system(2, 0, P).lane_invariant) ? 1 : 0;
        } catch (const std::exception &e) {
            tlab_set_error(e.what());
            return TLAB_EINVAL;
        }
    }
    switch (what) {
    case 0: return p->t.n;
    case 1: return p->t.der1.ndl;
    case 2: return p->t.der1.ndr;
    case 3: return p->t.der2.ndl;
    case 4: return p->t.der2.ndr;
    case 5: return p->t.der2.need_1der ? 1 : 0;
    case 6: return p->t.periodic ? 1 : 0;
    case 7: return p->t.stagger ? 1 : 0;
    }
    return TLAB_EINVAL;
}

int tlab_fdm_plan_get(tlab_fdm_plan_t p, int which, double *buf, int nbuf) {
    if (!p || !buf) return TLAB_EINVAL;
    const FdmTables &t = p->t;
    const double *src = nullptr;
    size_t m = 0;
    switch (which) {
    case 1: src = t.der1.lhs.data(); m = t.der1.lhs.size(); break;
    case 2: src = t.der1.rhs.data(); m = t.der1.rhs.size(); break;
    case 3: src = t.der1.lu.data(); m = t.der1.lu.size(); break;
    case 4: src = t.der1.rhs_b; m = 32; break;
    case 5: src = t.der1.rhs_t; m = 35; break;
    case 6: src = t.der1.mwn.data(); m = t.der1.mwn.size(); break;
    case 7: src = t.der2.lhs.data(); m = t.der2.lhs.size(); break;
    case 8: src = t.der2.rhs.data(); m = t.der2.rhs.size(); break;
    case 9: src = t.der2.lu.data(); m = t.der2.lu.size(); break;
    case 10: src = t.der2.mwn.data(); m = t.der2.mwn.size(); break;
    case 11: src = t.jac.data(); m = t.jac.size(); break;
    case 12: src = t.lu0i.data(); m = t.lu0i.size(); break;
    case 13: src = t.lu1i.data(); m = t.lu1i.size(); break;
    default: return TLAB_EINVAL;
    }
    if ((size_t)nbuf < m) return TLAB_EINVAL;
    std::copy(src, src + m, buf);
    return (int)m;
}

// ------------------------------------------------------------------------------------------------
// operators
// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

LineGeom make_geom(int dir, int nx, int ny, int nz) {
    LineGeom g;
    const long long ntot = (long long)nx * ny * nz;
    if (dir == 1) { g.n = nx; g.row_stride = 1; g.lines_inner = 1; g.outer_stride = nx; }
    else if (dir == 2) { g.n = ny; g.row_stride = nx; g.lines_inner = nx; g.outer_stride = (long long)nx * ny; }
    else { g.n = nz; g.row_stride = (long long)nx * ny; g.lines_inner = nx * ny; g.outer_stride = 0; }
    g.nlines = ntot / g.n;
    return g;
}

enum { PATH_GENERIC = 1, PATH_XLINE = 2, PATH_RTILE = 3 };

// fusions used by the RHS driver (rhs.cpp): operand = in0 + scale*in0b (P1), result accumulated into the output (P1, Burgers)
struct OpExtra {
    const double *in0b = nullptr;
    double scale = 0.0;
    bool acc = false;
    // MODE_BURGERS with several transported fields and one advecting velocity
    int nf = 0;
    const double *fs[4] = {nullptr, nullptr, nullptr, nullptr};
    double *fo[4] = {nullptr, nullptr, nullptr, nullptr};
    double fnu[4] = {0, 0, 0, 0};
    // MODE_P1 final-update epilogue
    double *fq = nullptr;
    double fdte = 0.0, fkco = 1.0;
    int fscale = 0, fnx = 1, fny = 1;
    const double *fpb = nullptr, *fpt = nullptr;      // ... with given wall tendencies (Neumann walls) instead of zero
    int ffin[4] = {0, 0, 0, 0};
    double *fdiv = nullptr;
    double fidte = 0.0;
    unsigned fresh_mask = 0;        // k_htile MODE_BURGERS: fields that overwrite their tendency in an accumulating launch
    const double *ari = nullptr;    // MODE_BURGERS, anelastic: ribackground [ny] on the diffusion term
    int ari_mode = 0, ari_nx = 1, ari_ny = 1;
    bool sub = false;               // MODE_P1: out0 -= value
    int fneu = 0;                   // k_rtile MODE_P1: Neumann-final epilogue (RTileArgs::fneu)
    double fcb[4] = {0, 0, 0, 0}, fct[4] = {0, 0, 0, 0};
};
const OpExtra kNoExtra{};

// Periodic x lines on P chunks whose lane-variant tables the kernel keeps as float differences from chunk 0 (kernels.hip, xcoef): exact only
// while the chunks differ by rounding-level amounts (periodic "uniform" grids), which is checked here once per plan and P on the host tables.
bool xline_wide_ok(tlab_fdm_plan_t g, int P) {
    if (!g || !g->t.periodic || (P != 64 && P != 128 && P != 256) || g->t.n % P != 0) return false;
    int &cache = g->wide_ok[P == 64 ? 0 : P == 128 ? 1 : 2];
    if (cache >= 0) return cache != 0;
    bool ok = true;
    for (int which = 1; which <= 2 && ok; ++which) {
        const SystemEntry *ep = nullptr;
        try { ep = &g->system(which, 0, P); } catch (const std::exception &) { ok = false; break; }      // e.g. no two-level tables
        const SystemEntry &e = *ep;
        if (e.lane_invariant) continue;
        const ChunkedTables &h = e.host;
        const int m = h.m;
        const std::vector<double> *tabs[5] = {&h.Lm, &h.Dinv, &h.Cm, &h.V, &h.W};
        for (int t = 0; t < 5 && ok; ++t)
            for (int l = 0; l < P && ok; ++l)
                for (int p = 0; p < m; ++p) {
                    const double c = (*tabs[t])[(size_t)l * m + p], b = (*tabs[t])[p];
                    const volatile float df = (float)(c - b);
                    const volatile double r = b + (double)df;
                    if (r != c) { ok = false; break; }
                }
    }
    cache = ok ? 1 : 0;
    return ok;
}

// chunks per x line of the wave-per-line kernel (0: not on that kernel).  Lines of 1024 / 2048 periodic points go on 2 / 4 waves with 8 rows
// per lane (128 / 256 chunks) when their tables allow the float-difference form; TLAB_XLINE_WIDE=0 keeps the one-wave forms (16 / 32 rows per lane).
int xline_chunks(int n, tlab_fdm_plan_t g) {
    static const bool wide = [] { const char *e = getenv("TLAB_XLINE_WIDE"); return !(e && atoi(e) == 0); }();
    if (g && wide && n == 1024 && xline_wide_ok(g, 128)) return 128;
    if (g && wide && n == 2048 && xline_wide_ok(g, 256)) return 256;
    if (n == 2048) return (g && xline_wide_ok(g, 64)) ? 64 : 0;
    return xline_supported(n) ? 64 : 0;
}

int choose_path(int dir, int n, tlab_fdm_plan_t g = nullptr) {
    int path = PATH_GENERIC;
    if (g && g->t.der1.ndl == 5) return PATH_GENERIC;      // CompactJacobian6Penta: the first derivative runs on k_penta1, nothing is fused
    if (g && g->t.der1.direct && dir == 1) return PATH_GENERIC;      // per-row first-derivative RHS: not in the wave-per-line kernel
    if (dir == 1 && xline_chunks(n, g) > 0) path = PATH_XLINE;
    if (dir != 1 && (rtile_chunk(n) > 0 || htile_chunk(n, MODE_P1) > 0)) path = PATH_RTILE;
    if (g_force_path == PATH_GENERIC) path = PATH_GENERIC;
    if (g_force_path == PATH_RTILE && (rtile_chunk(n) > 0 || htile_chunk(n, MODE_P1) > 0) && dir != 1) path = PATH_RTILE;
    return path;
}

void run_generic(tlab_fdm_plan_t g, const LineGeom &geom, int which, int ibc, const double *in0, const double *d1in,
                 double *out) {
    if (which == 1 && g->t.der1.ndl == 5) {      // CompactJacobian6Penta (fdm_com1_jacobian.f90:136-192)
        const DerTables &d = g->t.der1;
        if (d.ndr != 7 || d.lu.empty()) throw Unsupported("pentadiagonal first derivative: 7 RHS diagonals and the LU factors are required");
        if (!g->penta_rhs) {
            g->penta_rhs = std::make_unique<DeviceArray>();
            g->penta_rhs->upload(d.rhs);
            g->penta_lu = std::make_unique<DeviceArray>();
            g->penta_lu->upload(d.lu);
        }
        PentaArgs a;
        a.in0 = in0; a.out0 = out; a.g = geom; a.rhs = g->penta_rhs->p; a.lu = g->penta_lu->p;
        a.periodic = d.periodic ? 1 : 0; a.ibc = d.periodic ? 0 : ibc;
        std::copy(d.rhs_b, d.rhs_b + 32, a.rb);
        std::copy(d.rhs_t, d.rhs_t + 35, a.rt);
        auto tile = [&](const LineGeom &tg, const double *src, double *dst, bool along_x = false) {      // k_pentatile where the line length allows (32-row chunks, at most 16)
            const int key = d.periodic ? 0 : ibc;
            auto &slot = g->penta_tile[key];
            if (!slot) {
                std::vector<double> rows, blocks, smw;
                pentatile_build(tg.n, tg.n / 32, d.periodic, key, d.lu.data(), rows, blocks, smw);
                slot = std::make_unique<tlab_fdm_plan::PentaTile>();
                slot->rows.upload(rows); slot->blocks.upload(blocks); slot->smw.upload(smw);
            }
            PentaTileArgs t;
            t.in0 = src; t.out0 = dst; t.g = tg; t.rhs = g->penta_rhs->p; t.rows = slot->rows.p; t.blocks = slot->blocks.p; t.smw = slot->smw.p;
            t.r6 = d.rhs[4 + (size_t)tg.n * 5]; t.r7 = d.rhs[4 + (size_t)tg.n * 6];      // rhs(5, 6), rhs(5, 7)
            t.periodic = a.periodic; t.ibc = a.ibc;
            std::copy(d.rhs_b, d.rhs_b + 32, t.rb);
            std::copy(d.rhs_t, d.rhs_t + 35, t.rt);
            if (along_x) hip_check(launch_pentatile_x(t, g_stream), "k_pentatile<x>");
            else hip_check(launch_pentatile(t, g_stream), "k_pentatile");
        };
        if (geom.row_stride == 1 && pentatile_x_ok(geom)) {      // x lines of 64 .. 512 points: the tile kernel through an LDS tile, no transposes
            tile(geom, in0, out, true);
            return;
        }
        if (geom.row_stride == 1 && geom.nlines >= 64) {
            // x lines: one thread per line strides through contiguous memory (64 cache lines per wave access; 243 GB/s at 256^3, measured).  Like the
            // reference (OPR_Partial_X: TLab_Transpose, solve, transpose back, opr_partial.f90:185-195) the lines are made the fastest index first:
            // two transposes at the copy rate + the coalesced solve (the y-direction speed) -- same operations per line, same results
            const size_t N = (size_t)geom.n * geom.nlines;
            if (!g->penta_ws) g->penta_ws = std::make_unique<DeviceArray>();
            if (g->penta_ws->n < 2 * N) g->penta_ws->alloc(2 * N);
            double *t1 = g->penta_ws->p, *t2 = t1 + N;
            hip_check(launch_transpose(in0, t1, geom.n, (int)geom.nlines, g_stream), "transpose");
            a.in0 = t1; a.out0 = t2;
            a.g.row_stride = geom.nlines; a.g.lines_inner = (int)geom.nlines; a.g.outer_stride = 0;
            if (pentatile_ok(a.g)) tile(a.g, t1, t2);
            else hip_check(launch_penta1(a, g_stream), "k_penta1");
            hip_check(launch_transpose(t2, out, (int)geom.nlines, geom.n, g_stream), "transpose");
            return;
        }
        if (pentatile_ok(geom)) tile(geom, in0, out);
        else hip_check(launch_penta1(a, g_stream), "k_penta1");
        return;
    }
    GenericArgs a;
    a.in0 = in0; a.in1 = d1in; a.out0 = out; a.g = geom;
    a.s = g->stencil(which, ibc);
    a.y = g->system(which, ibc, 1).dev();
    a.jc = (which == 2 && d1in) ? g->jaccorr() : JacCorrDev{nullptr};
    hip_check(launch_generic(which == 2, a, g_stream), "k_generic");
}

void run_rtile(tlab_fdm_plan_t g, const LineGeom &geom, int mode, int ibc, const double *in0, const double *in1,
               const double *in2, double *out, double nu, const OpExtra &ex = kNoExtra) {
    const int P = geom.n / rtile_chunk(geom.n);
    RTileArgs a{};
    a.in0 = in0; a.in1 = in1; a.in2 = in2; a.out0 = out; a.out1 = nullptr; a.g = geom; a.nu = nu;
    a.in0b = ex.in0b; a.in0b_scale = ex.scale; a.acc = ex.sub ? 2 : (ex.acc ? 1 : 0);
    a.nf = 0;
    a.fq = ex.fq; a.fdte = ex.fdte; a.fkco = ex.fkco; a.fscale = ex.fscale; a.fnx = ex.fnx; a.fny = ex.fny; a.fpb = ex.fpb; a.fpt = ex.fpt;
    a.fneu = ex.fneu;
    for (int q = 0; q < 4; ++q) { a.fcb[q] = ex.fcb[q]; a.fct[q] = ex.fct[q]; }
    a.s1 = g->stencil(1, ibc);
    a.s2 = g->stencil(2, 0);
    a.y1 = g->system(1, ibc, P).dev();
    a.y2 = g->system(2, 0, P).dev();
    a.jc = (mode == MODE_P2_D1IN || mode == MODE_BURGERS_D1IN) ? g->jaccorr() : JacCorrDev{nullptr};
    hip_check(launch_rtile(mode, a, g_stream), "k_rtile");
}

int g_htile_policy = 0;   // 0 automatic, 1 never (two-launch k_rtile path), 2 always when the size allows

bool htile_ok(int n, int mode) { return g_htile_policy != 1 && htile_chunk(n, mode) > 0; }

void run_htile(tlab_fdm_plan_t g, const LineGeom &geom, int mode, int ibc, const double *in0, const double *vel, double *out0,
               double *out1, double nu, const OpExtra &ex = kNoExtra) {
    const int C = geom.n / htile_chunk(geom.n, mode);
    RTileArgs a{};
    a.in0 = in0; a.in1 = nullptr; a.in2 = vel; a.out0 = out0; a.out1 = out1; a.g = geom; a.nu = nu;
    a.in0b = nullptr; a.in0b_scale = 0.0; a.acc = ex.acc ? 1 : 0;
    a.fq = nullptr; a.fdte = 0.0; a.fkco = 1.0; a.fscale = 0; a.fnx = 1; a.fny = 1; a.fpb = nullptr; a.fpt = nullptr;
    if (mode == MODE_P1) {      // the epilogues of k_rtile<P1> (run_p1_tile)
        a.in0b = ex.in0b; a.in0b_scale = ex.scale; a.acc = ex.sub ? 2 : (ex.acc ? 1 : 0);
        a.fq = ex.fq; a.fdte = ex.fdte; a.fkco = ex.fkco; a.fscale = ex.fscale; a.fnx = ex.fnx; a.fny = ex.fny; a.fpb = ex.fpb; a.fpt = ex.fpt;
    }
    a.nf = ex.nf > 0 ? ex.nf : 1;
    for (int f = 0; f < 4; ++f) { a.fs[f] = ex.nf > 0 ? ex.fs[f] : in0; a.fo[f] = ex.nf > 0 ? ex.fo[f] : out0; a.fnu[f] = ex.nf > 0 ? ex.fnu[f] : nu; }
    a.s1 = g->stencil(1, ibc);
    a.s2 = g->stencil(2, 0);
    a.y1 = g->system(1, ibc, C).dev();
    a.y2 = g->system(2, 0, C).dev();
    a.jc = (mode != MODE_P1) ? g->jaccorr() : JacCorrDev{nullptr};
    a.fresh_mask = ex.fresh_mask; a.fdiv = ex.fdiv; a.fidte = ex.fidte;
    a.ari = ex.ari; a.ari_mode = ex.ari_mode; a.ari_nx = ex.ari_nx; a.ari_ny = ex.ari_ny;
    hip_check(launch_htile(mode, a, g_stream), "k_htile");
}

// One line-set (first derivative, with or without the fused epilogues) along y or z.  k_rtile's 64-line tiles are the best form while a chunk of
// the line fits the 128 VGPRs of its 1024-thread launch (32 rows, lines up to 512 points); at 1024 points its chunks are 64 rows and it spills
// 431 VGPRs (3.0 TB/s for the final z gradient of BASELINE configs[3]) -- those lines take k_htile's 32-line tiles (32 chunks of 32 rows, 126 VGPRs),
// which carries the same epilogues except the Neumann-final one (y lines only).
void run_p1_tile(tlab_fdm_plan_t g, const LineGeom &geom, int ibc, const double *u, double *result, const OpExtra &ex = kNoExtra) {
    static const bool off = [] { const char *e = getenv("TLAB_P1_HTILE"); return e && atoi(e) == 0; }();
    // ... and the final-update epilogue (five passes: p, h, q in; h, q out) at every length: k_rtile's 1024-thread launch spills 43 VGPRs with it,
    // k_htile<32, P1, 512> none -- final z gradient at 512^3 1.24 -> 1.15 ms (A/B on one box); TLAB_P1_HTILE=2: k_htile for every epilogue form
    static const bool always = [] { const char *e = getenv("TLAB_P1_HTILE"); return e && atoi(e) == 2; }();
    const int mr = rtile_chunk(geom.n);
    const bool rtile_spills = mr == 64 && geom.n / 64 > 8;
    const bool lane_offsets_fit = 3.0 * 32.0 * (double)geom.row_stride * 8.0 + 512.0 < 4294967296.0;      // launch_htile's 32-bit lane part of an address
    // ... and y lines (row stride = lines per plane): plain 0.509 -> 0.49 ms at 512^3, with the operand sum of the forcing term 0.114 -> 0.101 ms per 64-plane
    // slab; plain z lines stay on k_rtile (0.536 against 0.585 ms)
    const bool ylines = geom.row_stride == geom.lines_inner && geom.lines_inner != geom.nlines;
    if (!off && (rtile_spills || always || ex.fq != nullptr || ylines) && ex.fneu == 0 && htile_chunk(geom.n, MODE_P1) == 32 && g_htile_policy != 1 && lane_offsets_fit)
        run_htile(g, geom, MODE_P1, ibc, u, nullptr, result, nullptr, 0.0, ex);
    else
        run_rtile(g, geom, MODE_P1, ibc, u, nullptr, nullptr, result, 0.0, ex);
}

void run_xline(tlab_fdm_plan_t g, const LineGeom &geom, int mode, int ibc, const double *in0, const double *in1,
               double *out0, double *out1, double nu, const OpExtra &ex = kNoExtra) {
    XLineArgs a;
    a.in0 = in0; a.in1 = in1; a.out0 = out0; a.out1 = out1; a.nlines = geom.nlines; a.nu = nu;
    a.in0b = ex.in0b; a.in0b_scale = ex.scale; a.acc = ex.sub ? 2 : (ex.acc ? 1 : 0);
    a.fq = ex.fq; a.fdte = ex.fdte; a.fkco = ex.fkco; a.fscale = ex.fscale; a.fnx = ex.fnx; a.fny = ex.fny; a.fpb = ex.fpb; a.fpt = ex.fpt;
    a.nf = ex.nf > 0 ? ex.nf : 1;
    for (int f = 0; f < 4; ++f) { a.fs[f] = ex.nf > 0 ? ex.fs[f] : in0; a.fo[f] = ex.nf > 0 ? ex.fo[f] : out0; a.fnu[f] = ex.nf > 0 ? ex.fnu[f] : nu; a.ffin[f] = ex.ffin[f]; }
    a.fdiv = ex.fdiv; a.fidte = ex.fidte;
    a.ari = ex.ari; a.ari_ny = ex.ari_ny;
    a.s1 = g->stencil(1, ibc);
    a.s2 = g->stencil(2, 0);
    const int P = xline_chunks(geom.n, g);
    SystemEntry &e1 = g->system(1, ibc, P), &e2 = g->system(2, 0, P);
    a.y1 = e1.dev();
    a.y2 = e2.dev();
    const bool lv = !(e1.lane_invariant && e2.lane_invariant);
    static const bool dbg = getenv("TLAB_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[tlab] k_xline mode %d n %d chunks %d lane_variant %d\n", mode, geom.n, P, (int)lv);
    hip_check(launch_xline(mode, geom.n, P, lv, a, g_stream), "k_xline");
}

void check_common(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc) {
    if (!g) throw Invalid("null plan");
    if (dir < 1 || dir > 3) throw Invalid("dir must be 1, 2 or 3");
    if (nx < 1 || ny < 1 || nz < 1) throw Invalid("bad sizes");
    if ((long long)nx * ny * nz > 2147483647LL) throw Invalid("local field exceeds int32 indexing (reference limit, tlab_memory.f90:175)");
    if (ibc < 0 || ibc > 3) throw Invalid("ibc must be 0..3");
    const int n = (dir == 1) ? nx : (dir == 2) ? ny : nz;
    if (g->t.n != n) throw Invalid("plan size does not match the field size along dir");
    if (g_device < 0) throw HipError("tlab_init has not been called (no CPU fallback exists)");
}

}  // namespace

// ---- internal fused variants for the RHS driver; return false when the sizes are not on a fused fast path ----
// result (+)= d/dx_dir (u + scale*ub)
bool tlab_internal_partial_p1_fusable(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz) {
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    const int path = choose_path(dir, geom.n, g);    // with the plan: x lines of 2048 points take k_xline when their tables allow it
    return path == PATH_XLINE || (path == PATH_RTILE && rtile_chunk(geom.n) > 0);
}
bool tlab_internal_partial_p1_fused(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, const double *u, const double *ub,
                                    double scale, double *result, bool acc) {
    check_common(dir, g, nx, ny, nz, ibc);
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    OpExtra ex;
    ex.in0b = ub; ex.scale = scale; ex.acc = acc;
    const int path = choose_path(dir, geom.n, g);
    if (path == PATH_XLINE) {
        run_xline(g, geom, MODE_P1, ibc, u, nullptr, result, nullptr, 0.0, ex);
    } else if (path == PATH_RTILE && rtile_chunk(geom.n) > 0) {
        run_p1_tile(g, geom, ibc, u, result, ex);
    } else {
        return false;
    }
    g_last_path = path;
    return true;
}
// coefficients of BOUNDARY_BCS_NEUMANN_Y (boundary_bcs.f90:368-473): wall value = (u(2) cb0 + u(3) cb1) + u(4) cb2 + cb3 du(2), likewise at the top
void neumann_y_coefficients(tlab_fdm_plan_t g, int ibc, int ny, double (&cb)[4], double (&ct)[4]) {
    const DerTables &d = g->t.der1;
    if (d.ndl != 3 || (d.ndr != 3 && d.ndr != 5)) throw Unsupported("BOUNDARY_BCS_NEUMANN_Y: tridiagonal first-derivative schemes only");
    if (ny < 8) throw Invalid("BOUNDARY_BCS_NEUMANN_Y: ny too small");
    std::vector<double> lhs(d.lhs.begin(), d.lhs.begin() + (size_t)3 * ny);
    double rb[4 * 8] = {0}, rt[5 * 7] = {0};
    fdm_bcs_neumann(ibc, ny, 3, lhs.data(), d.ndr, d.rhs.data(), rb, rt);
    for (int q = 0; q < 4; ++q) cb[q] = ct[q] = 0.0;
#define RB(j, c) rb[((j)-1) + 4 * (c)]
#define RT(rr, c) rt[(rr) + 5 * ((c)-1)]
    if (d.ndr == 5) {   // MatMul_5d_antisym, fdm_matmul.f90:384,410
        cb[0] = RB(1, 4); cb[1] = RB(1, 5); cb[2] = RB(1, 1);
        ct[0] = RT(3, 5); ct[1] = RT(3, 1); ct[2] = RT(3, 2);
    } else {            // MatMul_3d_antisym, fdm_matmul.f90:179,203
        cb[0] = RB(1, 3); cb[1] = RB(1, 1);
        ct[1] = RT(2, 3); ct[2] = RT(2, 1);
    }
#undef RB
#undef RT
    cb[3] = lhs[0 + (size_t)ny * 2];          // lu(1, ip+idl+1): first row of the reduced LHS is not touched by TRIDFS
    ct[3] = lhs[(ny - 1) + (size_t)ny * 0];   // lu(ny, ip+idl-1)
}
// result -= d/dx_dir u  (fused kernels only; the caller falls back to OPR_Partial + a subtraction otherwise)
bool tlab_internal_partial_p1_sub(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, const double *u, double *result) {
    check_common(dir, g, nx, ny, nz, 0);
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    OpExtra ex;
    ex.sub = true;
    const int path = choose_path(dir, geom.n, g);
    if (path == PATH_XLINE) run_xline(g, geom, MODE_P1, 0, u, nullptr, result, nullptr, 0.0, ex);
    else if (path == PATH_RTILE && rtile_chunk(geom.n) > 0) run_p1_tile(g, geom, 0, u, result, ex);
    else return false;
    g_last_path = path;
    return true;
}
// The last pass over a field with Neumann walls (ibc: 1 jmin, 2 jmax, 3 both; the other side Dirichlet): BOUNDARY_BCS_NEUMANN_Y on the finished
// tendency h, its wall planes, q += dte h, h *= kco -- one launch along y instead of OPR_Partial_Y + k_neumann_planes + k_final_update.
bool tlab_internal_neumann_final_ok(tlab_fdm_plan_t g, int nx, int ny, int nz) {
    static const bool on = [] { const char *e = getenv("TLAB_NEUMANN_FINAL"); return !(e && atoi(e) == 0); }();
    if (!on || !g || g->t.periodic || g->t.der1.direct || ny < 8) return false;
    const DerTables &d = g->t.der1;
    if (d.ndl != 3 || (d.ndr != 3 && d.ndr != 5)) return false;
    const LineGeom geom = make_geom(2, nx, ny, nz);
    const int M = rtile_chunk(geom.n);
    return choose_path(2, geom.n, g) == PATH_RTILE && M >= 8 && M % 8 == 0;
}
bool tlab_internal_neumann_final(tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, double *h, double *q, double dte, double kco, int scale) {
    check_common(2, g, nx, ny, nz, ibc);
    if (ibc < 1 || ibc > 3 || !tlab_internal_neumann_final_ok(g, nx, ny, nz)) return false;
    const LineGeom geom = make_geom(2, nx, ny, nz);
    OpExtra ex;
    ex.fq = q; ex.fdte = dte; ex.fkco = kco; ex.fscale = scale; ex.fnx = nx; ex.fny = ny;
    ex.fneu = ibc;
    neumann_y_coefficients(g, ibc, ny, ex.fcb, ex.fct);
    run_rtile(g, geom, MODE_P1, ibc, h, nullptr, nullptr, h, 0.0, ex);
    g_last_path = PATH_RTILE;
    return true;
}
// h -= d/dx_dir p ; walls ; q += dte h ; h *= kco : the last pass over a velocity component folded into the gradient of the pressure.
// Dirichlet walls only (the tendency is zero on the wall planes); dir = 1 or 3.
// pb, pt (device, [nx][nz]; NULL: zero): the tendencies of the wall planes j = 0 / ny-1 (BOUNDARY_BCS_NEUMANN_Y's values for a Neumann wall)
bool tlab_internal_gradient_final(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, const double *p, double *q, double *h, double dte,
                                  double kco, int scale, const double *pb, const double *pt) {
    check_common(dir, g, nx, ny, nz, 0);
    if (dir == 2) return false;
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    OpExtra ex;
    ex.fq = q; ex.fdte = dte; ex.fkco = kco; ex.fscale = scale; ex.fnx = nx; ex.fny = ny; ex.fpb = pb; ex.fpt = pt;
    const int path = choose_path(dir, geom.n, g);
    if (path == PATH_XLINE) run_xline(g, geom, MODE_P1, 0, p, nullptr, h, nullptr, 0.0, ex);
    else if (path == PATH_RTILE && rtile_chunk(geom.n) > 0) run_p1_tile(g, geom, 0, p, h, ex);
    else return false;
    g_last_path = path;
    return true;
}
extern "C" bool tlab_internal_anelastic();      // (defined inside the extern "C" block below)
extern "C" bool tlab_internal_dealiasing();
// result += nu d2s - vel ds   (only when the fully fused Burgers kernels apply)
bool tlab_internal_burgers_acc(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, double nu, const double *s, const double *vel,
                               double *result) {
    check_common(dir, g, nx, ny, nz, ibc);
    if (tlab_internal_anelastic() || tlab_internal_dealiasing()) return false;      // those branches of OPR_Burgers_1D: the caller's unfused sequence
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    OpExtra ex;
    ex.acc = true;
    const bool corr = g->t.der2.need_1der || g->t.der2.direct;      // (the x-line kernel only knows the constant stencils)
    const int path = choose_path(dir, geom.n, g);
    if (path == PATH_XLINE && !corr) {
        run_xline(g, geom, MODE_BURGERS, ibc, s, vel, result, nullptr, nu, ex);
    } else if (path == PATH_RTILE && htile_ok(geom.n, MODE_BURGERS)) {
        run_htile(g, geom, MODE_BURGERS, ibc, s, vel, result, nullptr, nu, ex);
    } else {
        return false;
    }
    g_last_path = path;
    return true;
}

bool tlab_internal_burgers_fusable(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz) {
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    const int path = choose_path(dir, geom.n, g);
    return (path == PATH_XLINE && !g->t.der2.need_1der && !g->t.der2.direct) || (path == PATH_RTILE && htile_ok(geom.n, MODE_BURGERS));
}
// ... with the anelastic diffusion weight (tlab_internal_burgers_acc_n's ari): it exists in the wave-per-line kernel and in the 32-line tile form of k_htile
bool tlab_internal_burgers_fusable_anelastic(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz) {
    static const bool on = [] { const char *e = getenv("TLAB_ANELASTIC_FUSED"); return !(e && atoi(e) == 0); }();
    if (!on || !tlab_internal_burgers_fusable(dir, g, nx, ny, nz)) return false;
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (choose_path(dir, geom.n, g) == PATH_XLINE) return true;
    return htile_chunk(geom.n, MODE_BURGERS) == 32 && geom.n / 32 <= 16 && !htile_narrow() && (dir == 2 || nx % 32 == 0);
}
// several transported fields, one advecting velocity: result[f] += nu[f] d2 s[f] - vel d s[f]
// x lines of at most 512 points on the wave-per-line kernel: the launch can finish the substep of a transported field in its epilogue
bool tlab_internal_burgers_can_finish(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz) {
    if (dir != 1 || nx / std::max(xline_chunks(nx, g), 1) > 8) return false;      // the epilogue exists in the 8-rows-per-lane forms
    return tlab_internal_burgers_fusable(dir, g, nx, ny, nz) && choose_path(1, nx, g) == PATH_XLINE;
}

// finish (may be NULL): per field, != 0 -> this launch is the last term of that field's tendency and also does its Runge-Kutta update
// (s += dte h, h = scale ? kco h : h, h = 0 on the wall planes); only where tlab_internal_burgers_can_finish says so
// the y / z tile kernel can add its direction's term of the pressure forcing in the epilogue of a one-field launch (RTileArgs::fdiv)
bool tlab_internal_burgers_can_div(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz) {
    if (!g || dir < 2 || dir > 3) return false;
    static const bool on = [] { const char *e = getenv("TLAB_DIV_IN_BURGERS"); return !(e && atoi(e) == 0); }();
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (!on || geom.n == 1 || choose_path(dir, geom.n, g) != PATH_RTILE || !htile_ok(geom.n, MODE_BURGERS)) return false;
    return htile_chunk(geom.n, MODE_BURGERS) == 32 && geom.n / 32 <= 32 && !g->t.der1.direct && !htile_narrow();      // 32 chunks: 1024-point lines on 16-line tiles
}

bool tlab_internal_burgers_acc_n(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, int nf, const double *nu, const double *const *s,
                                 const double *vel, double *const *result, bool overwrite, const int *finish, double dte, double kco,
                                 int scale, double *divx, double idte, unsigned fresh_mask, const double *ari) {
    check_common(dir, g, nx, ny, nz, ibc);
    if (nf < 1 || nf > 4) throw Invalid("1 to 4 fields per call");
    if (tlab_internal_anelastic() && !ari) return false;      // the operator state says anelastic: the plain fused kernels would drop the density weight
    const LineGeom geom = make_geom(dir, nx, ny, nz);
    if (geom.n == 1) return false;
    OpExtra ex;
    ex.acc = !overwrite;        // overwrite: the tendency is known to be zero (start of a Runge-Kutta step): neither zero-filled nor read
    ex.fresh_mask = fresh_mask; // ... or only that of some fields
    if (tlab_internal_dealiasing()) return false;
    if (ari) {      // anelastic: ribackground [ny] (device) on the diffusion term; the epilogues are forms of the incompressible driver
        if (finish || divx || !tlab_internal_burgers_fusable_anelastic(dir, g, nx, ny, nz)) return false;
        ex.ari = ari; ex.ari_mode = dir == 2 ? 1 : 2; ex.ari_nx = nx; ex.ari_ny = ny;
    }
    ex.nf = nf;
    for (int f = 0; f < nf; ++f) { ex.fs[f] = s[f]; ex.fo[f] = result[f]; ex.fnu[f] = nu[f]; }
    if (finish) {
        if (!tlab_internal_burgers_can_finish(dir, g, nx, ny, nz)) throw Invalid("internal: this Burgers launch cannot finish a field");
        for (int f = 0; f < nf; ++f) ex.ffin[f] = finish[f];
        ex.fdte = dte; ex.fkco = kco; ex.fscale = scale; ex.fny = ny;
    }
    if (divx) {      // divx: the launch also writes (x) / adds (y, z) d/dx (h + idte vel) of the field that is the velocity itself (a term of the pressure forcing)
        if (!(dir == 1 ? tlab_internal_burgers_can_finish(dir, g, nx, ny, nz) : tlab_internal_burgers_can_div(dir, g, nx, ny, nz)))
            throw Invalid("internal: this Burgers launch cannot write the forcing term");
        ex.fdiv = divx; ex.fidte = idte;
    }
    const bool corr = g->t.der2.need_1der || g->t.der2.direct;
    const int path = choose_path(dir, geom.n, g);
    if (path == PATH_XLINE && !corr) {
        if (fresh_mask) throw Invalid("internal: per-field overwrite is a feature of the y / z tile kernel");
        run_xline(g, geom, MODE_BURGERS, ibc, s[0], vel, result[0], nullptr, nu[0], ex);
    } else if (path == PATH_RTILE && htile_ok(geom.n, MODE_BURGERS)) {
        if (divx && (nf != 1 || s[0] != vel || !tlab_internal_burgers_can_div(dir, g, nx, ny, nz)))
            throw Invalid("internal: the forcing term rides on a one-field launch of the velocity component of that direction");
        run_htile(g, geom, MODE_BURGERS, ibc, s[0], vel, result[0], nullptr, nu[0], ex);
    } else {
        return false;
    }
    g_last_path = path;
    return true;
}

extern "C" {

int tlab_last_kernel_path(void) { return g_last_path; }
int tlab_force_kernel_path(int path) {
    g_force_path = path;
    return TLAB_OK;
}
int tlab_set_tuning(int key, int value) {
    if (key == 1) { rtile_force_chunk(value); return TLAB_OK; }
    if (key == 2) { g_htile_policy = value; return TLAB_OK; }
    if (key == 3 && (value == 16 || value == 32)) { htile_set_lines(value); return TLAB_OK; }
    if (key == 4 && value >= 0) { ptile_set_grid(value); return TLAB_OK; }      // persistent workgroups of k_ptile (0 = one per CU, rounded to a multiple of 8)
    g_err = "tlab_set_tuning: unknown key";
    return TLAB_EINVAL;
}

// Accumulating variants (no counterpart in the reference, which always goes through a temporary and a pointwise loop,
// rhs_global_incompressible_1.f90:106-112, :257-259): fused kernels when the sizes are on a fast path, the reference's sequence otherwise.
int tlab_opr_burgers_add(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, double nu, const double *s, const double *vel,
                         double *result, double *tmp1, double *tmp2) {
    return guarded([&] {
        check_common(dir, g, nx, ny, nz, ibc);
        if (!s || !vel || !result || result == s || result == vel) throw Invalid("tlab_opr_burgers_add: null or aliased arrays");
        if (tlab_internal_burgers_acc(dir, g, nx, ny, nz, ibc, nu, s, vel, result)) return;
        if (!tmp1 || !tmp2 || tmp1 == tmp2 || tmp1 == result || tmp2 == result) throw Invalid("tlab_opr_burgers_add: the unfused path needs tmp1, tmp2");
        const int rc = tlab_opr_burgers(dir, g, s == vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, nx, ny, nz, ibc, nu, s, vel, tmp1, tmp2, 0);
        if (rc != TLAB_OK) throw Invalid(std::string("tlab_opr_burgers_add: ") + g_err);
        hip_check(launch_add1(result, tmp1, (long long)nx * ny * nz, g_stream), "k_add1");
    });
}

int tlab_opr_burgers_add_n(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, int nf, const double *nu, const double *const *s,
                           const double *vel, double *const *result, double *tmp1, double *tmp2, int overwrite) {
    return guarded([&] {
        check_common(dir, g, nx, ny, nz, ibc);
        if (nf < 1 || nf > 4 || !nu || !s || !vel || !result) throw Invalid("tlab_opr_burgers_add_n: bad arguments (1 to 4 fields)");
        for (int f = 0; f < nf; ++f)
            if (!s[f] || !result[f] || result[f] == s[f] || result[f] == vel) throw Invalid("tlab_opr_burgers_add_n: null or aliased arrays");
        if (!tlab_internal_anelastic() &&
            tlab_internal_burgers_acc_n(dir, g, nx, ny, nz, ibc, nf, nu, s, vel, result, overwrite != 0, nullptr, 0.0, 1.0, 0, nullptr, 0.0, 0u, nullptr)) return;
        for (int f = 0; f < nf; ++f) {
            if (overwrite) hip_check(hipMemsetAsync(result[f], 0, (size_t)nx * ny * nz * sizeof(double), g_stream), "memset");
            const int rc = tlab_opr_burgers_add(dir, g, nx, ny, nz, ibc, nu[f], s[f], vel, result[f], tmp1, tmp2);
            if (rc != TLAB_OK) throw Invalid(std::string("tlab_opr_burgers_add_n: ") + g_err);
        }
    });
}

int tlab_opr_gradient_final(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, const double *p, double *q, double *h, double dte, double kco,
                            int scale, double *tmp1) {
    return guarded([&] {
        check_common(dir, g, nx, ny, nz, 0);
        if (!p || !q || !h || q == h || p == q || p == h) throw Invalid("tlab_opr_gradient_final: null or aliased arrays");
        if (tlab_internal_gradient_final(dir, g, nx, ny, nz, p, q, h, dte, kco, scale, nullptr, nullptr)) return;
        if (!tmp1 || tmp1 == p || tmp1 == q || tmp1 == h) throw Invalid("tlab_opr_gradient_final: the unfused path needs tmp1");
        const int rc = tlab_opr_partial(dir, g, TLAB_OPR_P1, nx, ny, nz, 0, p, tmp1, nullptr);
        if (rc != TLAB_OK) throw Invalid(std::string("tlab_opr_gradient_final: ") + g_err);
        hip_check(launch_final_update(q, h, tmp1, nullptr, nullptr, dte, kco, scale, nx, ny, nz, g_stream), "k_final_update");
    });
}

int tlab_opr_partial_add(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, const double *u, const double *ub, double scale,
                         double *result, int acc, double *tmp1, double *tmp2) {
    return guarded([&] {
        check_common(dir, g, nx, ny, nz, ibc);
        if (!u || !result || result == u || result == ub) throw Invalid("tlab_opr_partial_add: null or aliased arrays");
        if (tlab_internal_partial_p1_fused(dir, g, nx, ny, nz, ibc, u, ub, scale, result, acc != 0)) return;
        if (!tmp1 || !tmp2 || tmp1 == tmp2 || tmp1 == result || tmp2 == result) throw Invalid("tlab_opr_partial_add: the unfused path needs tmp1, tmp2");
        const long long n = (long long)nx * ny * nz;
        const double *operand = u;
        if (ub) {
            hip_check(launch_axpy1(tmp1, u, ub, scale, n, g_stream), "k_axpy1");
            operand = tmp1;
        }
        const int rc = tlab_opr_partial(dir, g, TLAB_OPR_P1, nx, ny, nz, ibc, operand, acc ? tmp2 : result, nullptr);
        if (rc != TLAB_OK) throw Invalid(std::string("tlab_opr_partial_add: ") + g_err);
        if (acc) hip_check(launch_add1(result, tmp2, n, g_stream), "k_add1");
    });
}

// BOUNDARY_BCS_NEUMANN_Y (tools/dns/boundary_bcs.f90:368-473): the reduced derivative (zero at the chosen walls) is the ordinary
// OPR_Partial_Y under ibc; the wall values follow from the first / last row of the compact scheme in a plane kernel.
int tlab_boundary_bcs_neumann_y(tlab_fdm_plan_t g, int ibc, int nx, int ny, int nz, const double *u, double *bcs_hb, double *bcs_ht,
                                double *tmp1) {
    return guarded([&] {
        check_common(2, g, nx, ny, nz, ibc);
        if (ibc != BCS_ND && ibc != BCS_DN && ibc != BCS_NN) throw Invalid("BOUNDARY_BCS_NEUMANN_Y: ibc must be 1 (jmin), 2 (jmax) or 3 (both)");
        if (g->t.periodic) throw Invalid("BOUNDARY_BCS_NEUMANN_Y: periodic direction");
        if (!u || !tmp1 || u == tmp1 || !bcs_hb || !bcs_ht) throw Invalid("BOUNDARY_BCS_NEUMANN_Y: null or aliased arrays");
        double cb[4], ct[4];
        neumann_y_coefficients(g, ibc, ny, cb, ct);
        const int rc = tlab_opr_partial(2, g, TLAB_OPR_P1, nx, ny, nz, ibc, u, tmp1, nullptr);
        if (rc != TLAB_OK) throw Invalid(std::string("BOUNDARY_BCS_NEUMANN_Y: ") + tlab_last_error());
        hip_check(launch_neumann_planes(u, tmp1, cb, ct, (ibc & 1), (ibc & 2), bcs_hb, bcs_ht, nx, ny, nz, g_stream), "k_neumann_planes");
    });
}

int tlab_opr_partial(int dir, tlab_fdm_plan_t g, int type, int nx, int ny, int nz, int ibc, const double *u,
                     double *result, double *tmp1) {
    return guarded([&] {
        check_common(dir, g, nx, ny, nz, ibc);
        if (!u || !result || u == result || (tmp1 && (tmp1 == u || tmp1 == result))) throw Invalid("u, result, tmp1 must be distinct");
        if (type >= TLAB_OPR_P1_INT_VP && type <= TLAB_OPR_P0_INT_PV) {
            // interpolatory operators of the staggered pressure grid (opr_partial.f90:110-120, :228-238 -> FDM_Interpol / FDM_Interpol_Der1,
            // fdm_interpolate.f90:98-160): a 4-point right-hand side and the periodic tridiagonal LU of g%intl -- the same shape as the periodic
            // compact FILTER (5-point per-row right-hand side + TRIDPSS), so they run as four such filter objects per plan (k_filter1d)
            if (dir == 2) throw Unsupported("interpolatory operators along y are not built (the RHS staggers x and z only)");
            if (!g->t.stagger) throw Invalid("the plan has no interpolation tables: tlab_fdm_plan_set_stagger");
            if ((dir == 1 ? nx : nz) == 1) throw Invalid("interpolatory operator along a direction of one point");
            const int k = type - TLAB_OPR_P1_INT_VP;      // 0 P1 VP, 1 P1 PV, 2 P0 VP, 3 P0 PV
            if (!g->interp[k]) {
                const int n = g->t.n;
                const double c0 = 1.0 / 15.0, c1 = 17.0 / 189.0;      // fdm_com0_jacobian.f90:61, :338
                const double st[4][5] = {{0.0, -c1, -1.0, 1.0, c1},       // (u(i+1) - u(i)) + c (u(i+2) - u(i-1))      :347
                                         {-c1, -1.0, 1.0, c1, 0.0},       // (u(i) - u(i-1)) + c (u(i+1) - u(i-2))      :380
                                         {0.0, c0, 1.0, 1.0, c0},         // u(i+1) + u(i) + c (u(i+2) + u(i-1))        :70
                                         {c0, 1.0, 1.0, c0, 0.0}};        // u(i) + u(i-1) + c (u(i+1) + u(i-2))        :102
                std::vector<double> tab((size_t)10 * n);
                for (int q = 0; q < 5; ++q)
                    for (int i = 0; i < n; ++i) tab[(size_t)q * n + i] = st[k][q];
                const std::vector<double> &lu = (k < 2) ? g->t.lu1i : g->t.lu0i;
                std::copy(lu.begin(), lu.end(), tab.begin() + (size_t)5 * n);
                ok_or_throw(tlab_filter_create(&g->interp[k], TLAB_FILTER_COMPACT, n, 1, 0, 0, 10, tab.data()));
            }
            tlab_internal_filter_1d(dir, g->interp[k], nx, ny, nz, u, result, g_stream);
            g_last_path = PATH_GENERIC;
            return;
        }
        if (type != TLAB_OPR_P1 && type != TLAB_OPR_P2 && type != TLAB_OPR_P2_P1)
            throw Unsupported("OPR_Partial type not built on the device (the IBM variants stay on the CPU path)");
        const long long ntot = (long long)nx * ny * nz;
        const LineGeom geom = make_geom(dir, nx, ny, nz);
        if (geom.n == 1) {  // 2-D guard (opr_partial.f90:175-177, :287-289)
            hip_check(launch_fill(result, 0.0, ntot, g_stream), "fill");
            if (type == TLAB_OPR_P2_P1 && tmp1) hip_check(launch_fill(tmp1, 0.0, ntot, g_stream), "fill");
            return;
        }
        const bool corr = g->t.der2.need_1der;
        if (type == TLAB_OPR_P2_P1 && !tmp1) throw Invalid("OPR_P2_P1 needs tmp1");
        if (type == TLAB_OPR_P2 && corr && !tmp1) throw Invalid("OPR_P2 on a non-uniform grid needs tmp1 (opr_partial.f90:96)");
        int path = choose_path(dir, geom.n, g);
        if (path == PATH_XLINE && (corr || g->t.der2.direct) && type != TLAB_OPR_P1) path = PATH_GENERIC;  // non-uniform / direct-scheme x: rare, generic kernel
        g_last_path = path;
        if (path == PATH_XLINE) {
            const int mode = (type == TLAB_OPR_P1) ? MODE_P1 : (type == TLAB_OPR_P2) ? MODE_P2 : MODE_P2_P1;
            run_xline(g, geom, mode, ibc, u, nullptr, result, tmp1, 0.0);
        } else if (path == PATH_RTILE) {
            const bool r64 = rtile_chunk(geom.n) > 0 && g_htile_policy != 2;     // 64-line tiles: best for one line-set
            if (type == TLAB_OPR_P1) {
                if (r64) run_p1_tile(g, geom, ibc, u, result);
                else run_htile(g, geom, MODE_P1, ibc, u, nullptr, result, nullptr, 0.0);
            } else if (type == TLAB_OPR_P2_P1 && htile_ok(geom.n, MODE_P2_P1)) {
                run_htile(g, geom, MODE_P2_P1, ibc, u, nullptr, result, tmp1, 0.0);              // fused, one load of u
            } else if (type == TLAB_OPR_P2 && ((corr && htile_ok(geom.n, MODE_P2)) || !r64)) {
                run_htile(g, geom, MODE_P2, ibc, u, nullptr, result, nullptr, 0.0);              // first derivative stays in registers
            } else {
                const bool need_d1 = corr || type == TLAB_OPR_P2_P1;
                if (need_d1) run_rtile(g, geom, MODE_P1, ibc, u, nullptr, nullptr, tmp1, 0.0);
                if (corr) run_rtile(g, geom, MODE_P2_D1IN, ibc, u, tmp1, nullptr, result, 0.0);
                else run_rtile(g, geom, MODE_P2, ibc, u, nullptr, nullptr, result, 0.0);
            }
        } else {
            if (type == TLAB_OPR_P1) {
                run_generic(g, geom, 1, ibc, u, nullptr, result);
            } else {
                const bool need_d1 = corr || type == TLAB_OPR_P2_P1;
                if (need_d1) run_generic(g, geom, 1, ibc, u, nullptr, tmp1);
                run_generic(g, geom, 2, 0, u, corr ? tmp1 : nullptr, result);
            }
        }
    });
}

// nse_eqns == DNS_EQNS_ANELASTIC: the state OPR_Burgers_Initialize keeps in module variables (rhoinv(1), rhoinv(3), the modified U factors of
// fdmDiffusion(2); physics/opr_burgers.f90:128-183)
static std::unique_ptr<DeviceArray> g_anelastic_ri;
static int g_anelastic_ny = 0;
// the host copies and a change counter: the RHS drivers (rhs.cpp) follow this state instead of keeping one of their own that could disagree with it
static std::vector<double> g_anelastic_rb_host, g_anelastic_ri_host;
static unsigned long g_anelastic_version = 0;
int tlab_opr_burgers_set_anelastic(int ny, const double *rbackground, const double *ribackground) {
    return guarded([&] {
        if (ny <= 0 || !rbackground || !ribackground) {      // back to the incompressible operator
            g_anelastic_ri.reset();
            g_anelastic_ny = 0;
            g_anelastic_rb_host.clear(); g_anelastic_ri_host.clear();
            ++g_anelastic_version;
            return;
        }
        if (g_device < 0) throw HipError("tlab_init has not been called (no CPU fallback exists)");
        // along y the reference scales U's inverse diagonal by ribackground(j) and its superdiagonal by rbackground(j+1): x'(j) = x(j) ribackground(j)
        // provided rbackground * ribackground = 1, which is what the profiles of Thermo_Anelastic are; anything else is not the anelastic operator
        for (int j = 0; j < ny; ++j)
            if (std::fabs(rbackground[j] * ribackground[j] - 1.0) > 1e-14) throw Invalid("ribackground must be 1 / rbackground");
        g_anelastic_ri = std::make_unique<DeviceArray>();
        g_anelastic_ri->upload(std::vector<double>(ribackground, ribackground + ny));
        g_anelastic_ny = ny;
        g_anelastic_rb_host.assign(rbackground, rbackground + ny);
        g_anelastic_ri_host.assign(ribackground, ribackground + ny);
        ++g_anelastic_version;
    });
}
bool tlab_internal_anelastic() { return g_anelastic_ny > 0; }
// ny of the profiles (0: incompressible), the host copies and the change counter
int tlab_internal_anelastic_state(const double **rb, const double **rib, unsigned long *version) {
    *rb = g_anelastic_rb_host.data(); *rib = g_anelastic_ri_host.data(); *version = g_anelastic_version;
    return g_anelastic_ny;
}

// [Dealiasing] (physics/opr_burgers.f90:33, 71, 118-125): Dealiasing(1:3), one filter per direction (NULL = DNS_FILTER_NONE); NOT owned
static tlab_filter_t g_dealias[3] = {nullptr, nullptr, nullptr};
static DeviceArray *g_wsd[2] = {nullptr, nullptr};      // wrkdea(:, 1:2) (:34, :124): filtered velocity and filtered ds/dx
int tlab_opr_burgers_set_dealiasing(int dir, tlab_filter_t f) {
    return guarded([&] {
        if (dir < 1 || dir > 3) throw Invalid("dir must be 1, 2 or 3");
        g_dealias[dir - 1] = f;
    });
}
bool tlab_internal_dealiasing() { return g_dealias[0] || g_dealias[1] || g_dealias[2]; }
// a filter that is being destroyed while still set as Dealiasing(dir) is taken out (tlab_filter_destroy): no dangling pointer in the Burgers operators
void tlab_internal_dealiasing_forget(tlab_filter_t f) {
    for (int i = 0; i < 3; ++i)
        if (g_dealias[i] == f) g_dealias[i] = nullptr;
}
static double *dealias_ws(int k, size_t n) {
    if (!g_wsd[k]) g_wsd[k] = new DeviceArray();
    if (g_wsd[k]->n < n) {
        if (g_wsd[k]->p) hip_check(hipFree(g_wsd[k]->p), "hipFree");
        g_wsd[k]->p = nullptr;
        hip_check(hipMalloc((void **)&g_wsd[k]->p, n * sizeof(double)), "hipMalloc(wrkdea)");
        g_wsd[k]->n = n;
    }
    return g_wsd[k]->p;
}

int tlab_opr_burgers(int dir, tlab_fdm_plan_t g, int ivel, int nx, int ny, int nz, int ibc, double nu, const double *s,
                     const double *u, double *result, double *tmp1, int write_transposed) {
    return guarded([&] {
        check_common(dir, g, nx, ny, nz, ibc);
        if (ivel != TLAB_OPR_B_SELF && ivel != TLAB_OPR_B_U_IN) throw Invalid("ivel must be OPR_B_SELF or OPR_B_U_IN");
        if (!s || !result || !tmp1 || s == result || tmp1 == s || tmp1 == result) throw Invalid("s, result, tmp1 must be distinct");
        const double *vel = (ivel == TLAB_OPR_B_SELF) ? s : u;
        if (!vel || vel == result || vel == tmp1) throw Invalid("velocity must not alias result / tmp1");
        const long long ntot = (long long)nx * ny * nz;
        const LineGeom geom = make_geom(dir, nx, ny, nz);
        if (geom.n == 1) {  // opr_burgers.f90:207-210
            hip_check(launch_fill(result, 0.0, ntot, g_stream), "fill");
            return;
        }
        const bool corr = g->t.der2.need_1der;
        const bool wt = write_transposed && ivel == TLAB_OPR_B_SELF && (dir == 1 || (dir == 2 && nz > 1));
        double *d1 = wt ? workspace((size_t)ntot) : tmp1;
        int path = choose_path(dir, geom.n, g);
        if (path == PATH_XLINE && (corr || g->t.der2.direct)) path = PATH_GENERIC;
        g_last_path = path;
        if (g_anelastic_ny > 0 || g_dealias[dir - 1]) {      // OPR_Burgers_1D with rhoinv (opr_burgers.f90:504-507) and / or dealiasing (:478-500):
            // the two derivatives unfused, then filter(u) and filter(ds/dx) if asked for, then the (weighted) sum
            if (g_anelastic_ny > 0 && g_anelastic_ny != ny) throw Invalid("anelastic profiles were given for another ny");
            ok_or_throw(tlab_opr_partial(dir, g, TLAB_OPR_P2_P1, nx, ny, nz, ibc, s, result, d1));
            const double *uf = vel, *dsf = d1;
            if (g_dealias[dir - 1]) {
                double *w1 = dealias_ws(0, (size_t)ntot), *w2 = dealias_ws(1, (size_t)ntot);
                tlab_internal_filter_1d(dir, g_dealias[dir - 1], nx, ny, nz, vel, w1, g_stream);
                tlab_internal_filter_1d(dir, g_dealias[dir - 1], nx, ny, nz, d1, w2, g_stream);
                uf = w1; dsf = w2;
            }
            if (g_anelastic_ny > 0)
                hip_check(launch_burgers_epilogue_anelastic(result, uf, dsf, nu, g_anelastic_ri->p, nx, ny, ntot, g_stream), "burgers epilogue (anelastic)");
            else
                hip_check(launch_burgers_epilogue(result, uf, dsf, nu, ntot, g_stream), "burgers epilogue");
        } else if (path == PATH_XLINE) {
            run_xline(g, geom, MODE_BURGERS, ibc, s, vel, result, nullptr, nu);
        } else if (path == PATH_RTILE && htile_ok(geom.n, MODE_BURGERS)) {
            run_htile(g, geom, MODE_BURGERS, ibc, s, vel, result, nullptr, nu);                  // fully fused: s and vel read once
        } else if (path == PATH_RTILE) {
            run_rtile(g, geom, MODE_P1, ibc, s, nullptr, nullptr, d1, 0.0);
            run_rtile(g, geom, MODE_BURGERS_D1IN, ibc, s, d1, vel, result, nu);
        } else {
            run_generic(g, geom, 1, ibc, s, nullptr, d1);
            run_generic(g, geom, 2, 0, s, corr ? d1 : nullptr, result);
            hip_check(launch_burgers_epilogue(result, vel, d1, nu, ntot, g_stream), "burgers epilogue");
        }
        if (wt) {  // transposed operand exactly as the reference leaves it in tmp1 (opr_burgers.f90:250, :315)
            if (dir == 1) hip_check(launch_transpose(s, tmp1, nx, ny * nz, g_stream), "transpose");
            else hip_check(launch_transpose(s, tmp1, nx * ny, nz, g_stream), "transpose");
        }
    });
}

int tlab_transpose(const double *a, int nra, int nca, double *b) {
    return guarded([&] {
        if (!a || !b || a == b || nra < 1 || nca < 1) throw Invalid("tlab_transpose: bad arguments");
        if (g_device < 0) throw HipError("tlab_init has not been called");
        hip_check(launch_transpose(a, b, nra, nca, g_stream), "transpose");
    });
}

int tlab_debug_host_chunked_solve(tlab_fdm_plan_t p, int which, int ibc, int chunks, double *f) {
    return guarded([&] {
        if (!p || !f || (which != 1 && which != 2)) throw Invalid("bad arguments");
        TriDiag T = p->tridiag(which, (which == 2 || p->t.der1.periodic) ? 0 : ibc);
        ChunkedTables t;
        build_chunked(T, chunks, t);
        chunked_solve_host(t, f, chunks >= 64);
    });
}

}  // extern "C"
