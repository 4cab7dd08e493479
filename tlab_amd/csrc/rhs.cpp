// RHS_GLOBAL_INCOMPRESSIBLE_1 (tools/dns/rhs_global_incompressible_1.f90:15-405) and the explicit low-storage
// Runge-Kutta substep around it (TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT, tools/dns/time.f90:559-664, and the scaling
// of the tendencies, :261-298), orchestrating the device operators so that no field leaves HBM during a substep.
// Same operator sequence, same scratch roles (tmp1..tmp9 = txc(:,1:9)) as the reference.
#include "../../include/tlab_amd.h"

#include <hip/hip_runtime.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"

using namespace tlab;

extern hipStream_t tlab_current_stream();
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();

struct tlab_dns {
    tlab_fdm_plan_t g[3];
    tlab_poisson_plan_t poisson;
    int nx, ny, nz, nscal;
    double visc;
    std::vector<double> schmidt;
    double *bcs_hb = nullptr, *bcs_ht = nullptr;   // BcsFlowJmin%ref(:,:,2), BcsFlowJmax%ref(:,:,2)
    ~tlab_dns() {
        if (bcs_hb) (void)hipFree(bcs_hb);
        if (bcs_ht) (void)hipFree(bcs_ht);
    }
};

namespace {
struct Fail : std::runtime_error {
    int code;
    Fail(int c, const std::string &s) : std::runtime_error(s), code(c) {}
};
void ok(int rc, const char *what) {
    if (rc != TLAB_OK) throw Fail(rc, std::string(what) + ": " + tlab_last_error());
}
void hk(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Fail(TLAB_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace

extern "C" {

int tlab_dns_create(tlab_dns_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, tlab_poisson_plan_t poisson,
                    int nx, int ny, int nz, int nscal, double visc, const double *schmidt) {
    try {
        if (!out || !gx || !gy || !gz || !poisson || nscal < 0 || (nscal > 0 && !schmidt) || visc <= 0.0)
            throw Fail(TLAB_EINVAL, "tlab_dns_create: bad arguments");
        if (!tlab_device_ready()) throw Fail(TLAB_EHIP, "tlab_init has not been called (no CPU fallback exists)");
        auto d = std::make_unique<tlab_dns>();
        d->g[0] = gx; d->g[1] = gy; d->g[2] = gz;
        d->poisson = poisson;
        d->nx = nx; d->ny = ny; d->nz = nz; d->nscal = nscal; d->visc = visc;
        d->schmidt.assign(schmidt, schmidt + nscal);
        hk(hipMalloc((void **)&d->bcs_hb, (size_t)nx * nz * sizeof(double)), "hipMalloc");
        hk(hipMalloc((void **)&d->bcs_ht, (size_t)nx * nz * sizeof(double)), "hipMalloc");
        *out = d.release();
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

int tlab_dns_destroy(tlab_dns_t d) {
    delete d;
    return TLAB_OK;
}

int tlab_rhs_global_incompressible_1(tlab_dns_t d, double dte, double *const *q, double *const *s, double *const *hq,
                                     double *const *hs, double *const *txc) {
    try {
        if (!d || !q || !hq || !txc || (d->nscal > 0 && (!s || !hs)) || dte <= 0.0) throw Fail(TLAB_EINVAL, "tlab_rhs_global_incompressible_1: bad arguments");
        const int nx = d->nx, ny = d->ny, nz = d->nz;
        const long long n = (long long)nx * ny * nz;
        hipStream_t st = tlab_current_stream();
        double *u = q[0], *v = q[1], *w = q[2];
        double *tmp1 = txc[0], *tmp2 = txc[1], *tmp3 = txc[2], *tmp4 = txc[3], *tmp5 = txc[4], *tmp6 = txc[5], *tmp7 = txc[6],
               *tmp8 = txc[7], *tmp9 = txc[8];
        tlab_fdm_plan_t gx = d->g[0], gy = d->g[1], gz = d->g[2];
        const int B0 = 0;  // bcs = 0: biased, non-zero (:67)
        const double nu = d->visc;
        // ---- diffusion and advection (:98-136); the SELF calls leave their scratch in tmp4..6 ----
        ok(tlab_opr_burgers(1, gx, TLAB_OPR_B_SELF, nx, ny, nz, B0, nu, u, u, tmp1, tmp4, 0), "OPR_Burgers_X(u)");
        ok(tlab_opr_burgers(2, gy, TLAB_OPR_B_SELF, nx, ny, nz, B0, nu, v, v, tmp2, tmp5, 0), "OPR_Burgers_Y(v)");
        ok(tlab_opr_burgers(3, gz, TLAB_OPR_B_SELF, nx, ny, nz, B0, nu, w, w, tmp3, tmp6, 0), "OPR_Burgers_Z(w)");
        ok(tlab_opr_burgers(2, gy, TLAB_OPR_B_U_IN, nx, ny, nz, B0, nu, u, v, tmp7, tmp9, 0), "OPR_Burgers_Y(u)");
        ok(tlab_opr_burgers(3, gz, TLAB_OPR_B_U_IN, nx, ny, nz, B0, nu, u, w, tmp8, tmp9, 0), "OPR_Burgers_Z(u)");
        hk(launch_add3(hq[0], tmp1, tmp7, tmp8, n, st), "add3");
        ok(tlab_opr_burgers(1, gx, TLAB_OPR_B_U_IN, nx, ny, nz, B0, nu, v, u, tmp7, tmp9, 0), "OPR_Burgers_X(v)");
        ok(tlab_opr_burgers(3, gz, TLAB_OPR_B_U_IN, nx, ny, nz, B0, nu, v, w, tmp8, tmp9, 0), "OPR_Burgers_Z(v)");
        hk(launch_add3(hq[1], tmp2, tmp7, tmp8, n, st), "add3");
        ok(tlab_opr_burgers(1, gx, TLAB_OPR_B_U_IN, nx, ny, nz, B0, nu, w, u, tmp7, tmp9, 0), "OPR_Burgers_X(w)");
        ok(tlab_opr_burgers(2, gy, TLAB_OPR_B_U_IN, nx, ny, nz, B0, nu, w, v, tmp8, tmp9, 0), "OPR_Burgers_Y(w)");
        hk(launch_add3(hq[2], tmp3, tmp7, tmp8, n, st), "add3");
        // ---- scalars (:149-162) ----
        for (int is = 0; is < d->nscal; ++is) {
            const double kap = d->visc / d->schmidt[is];   // opr_burgers.f90:97
            ok(tlab_opr_burgers(1, gx, TLAB_OPR_B_U_IN, nx, ny, nz, B0, kap, s[is], u, tmp1, tmp9, 0), "OPR_Burgers_X(s)");
            ok(tlab_opr_burgers(2, gy, TLAB_OPR_B_U_IN, nx, ny, nz, B0, kap, s[is], v, tmp2, tmp9, 0), "OPR_Burgers_Y(s)");
            ok(tlab_opr_burgers(3, gz, TLAB_OPR_B_U_IN, nx, ny, nz, B0, kap, s[is], w, tmp3, tmp9, 0), "OPR_Burgers_Z(s)");
            hk(launch_add3(hs[is], tmp1, tmp2, tmp3, n, st), "add3");
        }
        // ---- pressure (:177-260, remove_divergence branch): forcing = div(hq + q/dte) ----
        hk(launch_axpy3(tmp2, tmp3, tmp4, hq[1], hq[0], hq[2], v, u, w, 1.0 / dte, n, st), "axpy3");
        ok(tlab_opr_partial(2, gy, TLAB_OPR_P1, nx, ny, nz, B0, tmp2, tmp1, nullptr), "OPR_Partial_Y");
        ok(tlab_opr_partial(1, gx, TLAB_OPR_P1, nx, ny, nz, B0, tmp3, tmp2, nullptr), "OPR_Partial_X");
        ok(tlab_opr_partial(3, gz, TLAB_OPR_P1, nx, ny, nz, B0, tmp4, tmp3, nullptr), "OPR_Partial_Z");
        hk(launch_sum3(tmp1, tmp2, tmp3, n, st), "sum3");
        // Neumann BCs in d/dy(p) s.t. v = 0 (:263-281)
        hk(launch_get_wall_planes(hq[1], d->bcs_hb, d->bcs_ht, nx, ny, nz, st), "wall planes");
        // pressure in tmp1, Oy derivative in tmp3 (:284)
        ok(tlab_opr_poisson(d->poisson, nx, ny, nz, TLAB_BCS_NN, tmp1, tmp2, tmp4, d->bcs_hb, d->bcs_ht, tmp3), "OPR_Poisson");
        ok(tlab_opr_partial(1, gx, TLAB_OPR_P1, nx, ny, nz, B0, tmp1, tmp2, nullptr), "OPR_Partial_X(p)");
        ok(tlab_opr_partial(3, gz, TLAB_OPR_P1, nx, ny, nz, B0, tmp1, tmp4, nullptr), "OPR_Partial_Z(p)");
        hk(launch_sub3(hq[0], hq[1], hq[2], tmp2, tmp3, tmp4, n, st), "sub3");
        // ---- boundary conditions (:360-398): no-slip walls / Dirichlet scalars -> tendencies vanish on the wall planes ----
        for (int iq = 0; iq < 3; ++iq) hk(launch_fill_wall_planes(hq[iq], 0.0, 0.0, nx, ny, nz, st), "wall planes");
        for (int is = 0; is < d->nscal; ++is) hk(launch_fill_wall_planes(hs[is], 0.0, 0.0, nx, ny, nz, st), "wall planes");
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

// ---- the pointwise pieces on their own, for drivers that interleave communication (z-slab decomposition) ----
#define PW_GUARD(expr)                                          \
    try {                                                       \
        if (!tlab_device_ready()) throw Fail(TLAB_EHIP, "tlab_init has not been called"); \
        hk((expr), "pointwise kernel");                         \
        return TLAB_OK;                                         \
    } catch (const Fail &f) {                                   \
        tlab_set_error(f.what());                               \
        return f.code;                                          \
    }
int tlab_pw_add3(double *h, const double *a, const double *b, const double *c, long long n) { PW_GUARD(launch_add3(h, a, b, c, n, tlab_current_stream())) }
int tlab_pw_axpy3(double *o1, double *o2, double *o3, const double *h1, const double *h2, const double *h3, const double *q1, const double *q2,
                  const double *q3, double s, long long n) { PW_GUARD(launch_axpy3(o1, o2, o3, h1, h2, h3, q1, q2, q3, s, n, tlab_current_stream())) }
int tlab_pw_sum3(double *a, const double *b, const double *c, long long n) { PW_GUARD(launch_sum3(a, b, c, n, tlab_current_stream())) }
int tlab_pw_sub3(double *h1, double *h2, double *h3, const double *a, const double *b, const double *c, long long n) { PW_GUARD(launch_sub3(h1, h2, h3, a, b, c, n, tlab_current_stream())) }
int tlab_pw_rk_update(double *q, double *h, double dte, double kco, int scale, long long n) { PW_GUARD(launch_rk_update(q, h, dte, kco, scale, n, tlab_current_stream())) }
int tlab_pw_get_wall_planes(const double *f, double *hb, double *ht, int nx, int ny, int nz) { PW_GUARD(launch_get_wall_planes(f, hb, ht, nx, ny, nz, tlab_current_stream())) }
int tlab_pw_fill_wall_planes(double *f, double vb, double vt, int nx, int ny, int nz) { PW_GUARD(launch_fill_wall_planes(f, vb, vt, nx, ny, nz, tlab_current_stream())) }

int tlab_time_substep_incompressible_explicit(tlab_dns_t d, double dte, double kco, int scale_tendencies, double *const *q,
                                              double *const *s, double *const *hq, double *const *hs, double *const *txc) {
    int rc = tlab_rhs_global_incompressible_1(d, dte, q, s, hq, hs, txc);
    if (rc != TLAB_OK) return rc;
    try {
        const long long n = (long long)d->nx * d->ny * d->nz;
        hipStream_t st = tlab_current_stream();
        for (int iq = 0; iq < 3; ++iq) hk(launch_rk_update(q[iq], hq[iq], dte, kco, scale_tendencies, n, st), "rk update");
        for (int is = 0; is < d->nscal; ++is) hk(launch_rk_update(s[is], hs[is], dte, kco, scale_tendencies, n, st), "rk update");
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

}  // extern "C"
