// RHS_GLOBAL_INCOMPRESSIBLE_1 (tools/dns/rhs_global_incompressible_1.f90:15-405) and the explicit low-storage
// Runge-Kutta substep around it (TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT, tools/dns/time.f90:559-664, and the scaling
// of the tendencies, :261-298), orchestrating the device operators so that no field leaves HBM during a substep.
// Same operator sequence, same scratch roles (tmp1..tmp9 = txc(:,1:9)) as the reference.
#include "../../include/tlab_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <functional>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"

using namespace tlab;

extern hipStream_t tlab_current_stream();
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();
int tlab_internal_deferred_flush();      // deferred.cpp: a recorded substep runs before the driver's state changes under it
bool tlab_internal_partial_p1_fused(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, const double *u, const double *ub,
                                    double scale, double *result, bool acc);
bool tlab_internal_partial_p1_fusable(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz);
bool tlab_internal_burgers_acc(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, double nu, const double *s, const double *vel,
                               double *result);
bool tlab_internal_gradient_final(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, const double *p, double *q, double *h, double dte,
                                  double kco, int scale, const double *pb = nullptr, const double *pt = nullptr);
bool tlab_internal_burgers_fusable(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz);
extern "C" bool tlab_internal_dealiasing();      // capi.cpp (defined inside its extern "C" block)
extern "C" int tlab_internal_anelastic_state(const double **rb, const double **rib, unsigned long *version);
bool tlab_internal_poisson_can_v_final(tlab_poisson_plan_t P);                                                      // poisson.hip
void tlab_internal_poisson_arm_v_final(tlab_poisson_plan_t P, double *q, double *h, double dte, double kco, int scale);
bool tlab_internal_burgers_can_finish(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz);
bool tlab_internal_burgers_acc_n(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, int nf, const double *nu, const double *const *s,
                                 const double *vel, double *const *result, bool overwrite, const int *finish, double dte, double kco, int scale,
                                 double *divx, double idte, unsigned fresh_mask = 0, const double *ari = nullptr);
bool tlab_internal_burgers_fusable_anelastic(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz);
bool tlab_internal_burgers_can_div(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz);
bool tlab_internal_partial_p1_sub(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, const double *u, double *result);
bool tlab_internal_neumann_final_ok(tlab_fdm_plan_t g, int nx, int ny, int nz);
bool tlab_internal_neumann_final(tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, double *h, double *q, double dte, double kco, int scale);

struct tlab_dns {
    tlab_fdm_plan_t g[3];
    tlab_poisson_plan_t poisson;
    int nx, ny, nz, nscal;
    double visc;
    std::vector<double> schmidt;
    double *bcs_hb = nullptr, *bcs_ht = nullptr;   // BcsFlowJmin%ref(:,:,2), BcsFlowJmax%ref(:,:,2)
    bool fuse = true;                              // fold the pointwise sums into the operator kernels where the fast kernels apply
    tlab_filter_t pfilter[3] = {nullptr, nullptr, nullptr};      // PressureFilter(1:3) (not owned) and its repeat counts
    int pfilter_rep[3] = {1, 1, 1};
    bool stagger = false;                          // [Staggering] StaggerHorizontalPressure: taken from the x / z plans at creation (tlab_fdm_plan_set_stagger)
    bool remove_divergence = true;                 // [Main] ... forcing = div(hq + q/dte) (rhs_global_incompressible_1.f90:177-232); false: div(hq) (:234-250)
    bool fresh = false;                            // one-shot: hq, hs count as zero on entry of the next substep (tlab_dns_begin_step)
    int flow_jmin[3] = {TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET};   // BcsFlowJmin%type
    int flow_jmax[3] = {TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET};
    std::vector<int> scal_jmin, scal_jmax;         // BcsScalJmin%type, BcsScalJmax%type
    // dynamic surface model of the scalars (BcsScalJmin/Jmax%SfcType = DNS_SFC_LINEAR, %cpl; boundary_bcs.f90:29-31, 49-50, 478-546)
    std::vector<int> sfc_jmin, sfc_jmax;
    std::vector<double> cpl_jmin, cpl_jmax;
    std::vector<double *> sref_b, sref_t;          // BcsScalJmin%ref(:,:,is), BcsScalJmax%ref(:,:,is) of the scalars with a surface model
    double *sfc_avg = nullptr;                     // one double: plane average
    // TIME_COURANT (tools/dns/time.f90:138-176): ds(ig)%one_ov_ds1 = 1/jac(:,1) on the device, dx2i, schmidtfactor
    double *od[3] = {nullptr, nullptr, nullptr};
    double *part = nullptr;                        // [2][NPART] partial (min, max) of the reductions
    double dx2i = 0.0, schmidtfactor = 0.0;
    int koffset = 0, nz_total = 0;                 // z-slab: first global plane and global nz (one_ov_ds1 of z is indexed globally)
    // nse_eqns == DNS_EQNS_ANELASTIC: rbackground, ribackground (ny values each) on the device; the wall values of rbackground on the host
    double *rb = nullptr, *rib = nullptr;
    // BOUNDARY_BCS_NEUMANN_Y as a weighted sum over the rows next to the wall (neumann_weights below): per variant ibc = 1..3 the device weights
    // [2][K] (bottom, top), K, and six scratch planes
    struct NeuW { double *w = nullptr; int K = 0; bool tried = false; } neuw[4];
    double *wall_planes = nullptr;
    double rb_wall[2] = {1.0, 1.0};
    unsigned long anel_version = 0;                // change counter of the operator state these mirror (follow_anelastic)
    bool anel_owner = false;                       // this driver switched the operator state on (tlab_dns_set_anelastic): it goes with the driver
    ~tlab_dns() {
        if (bcs_hb) (void)hipFree(bcs_hb);
        if (bcs_ht) (void)hipFree(bcs_ht);
        for (int i = 0; i < 3; ++i)
            if (od[i]) (void)hipFree(od[i]);
        if (part) (void)hipFree(part);
        for (double *p : sref_b) if (p) (void)hipFree(p);
        for (double *p : sref_t) if (p) (void)hipFree(p);
        if (sfc_avg) (void)hipFree(sfc_avg);
        if (rb) (void)hipFree(rb);
        if (rib) (void)hipFree(rib);
        for (NeuW &n : neuw) if (n.w) (void)hipFree(n.w);
        if (wall_planes) (void)hipFree(wall_planes);
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

// The wall value BOUNDARY_BCS_NEUMANN_Y (boundary_bcs.f90:368-473) gives a finished tendency h is a LINEAR functional of the y line:
// c0 h(2) + c1 h(3) + c2 h(4) + c3 (D_N h)(2) with D_N the first derivative under the Neumann variant of the system.  Its weights are found once per
// plan and variant by sending the unit vectors through the library's own routine (an ny x ny x 1 box holding the identity), and they decay like the
// coupling of the compact system (0.38^j): K = the rows beyond which they are below 1e-22 of the largest.  The wall planes of a field then cost a
// weighted sum over 2 K rows instead of a derivative pass over the field; and for u (w), whose finished tendency is hq - dp/dx (dp/dz), the sum
// commutes with the x (z) derivative: wall value = sum_j w_j hq(j) - d/dx [sum_j w_j p(j)] -- one derivative of a PLANE.
// (K <= ny always: on short lines the two sums overlap, which is as exact as the full line.)  Returns false where the weights are not finite.
bool neumann_weights(tlab_dns *d, int ibc) {
    tlab_dns::NeuW &W = d->neuw[ibc];
    if (W.tried) return W.w != nullptr;
    W.tried = true;
    const int ny = d->ny;
    const size_t N = (size_t)ny * ny;
    double *h = nullptr, *tmp = nullptr, *hb = nullptr, *ht = nullptr;
    auto release = [&] { for (double *p : {h, tmp, hb, ht}) if (p) (void)hipFree(p); };
    try {
        hk(hipMalloc((void **)&h, N * sizeof(double)), "hipMalloc");
        hk(hipMalloc((void **)&tmp, (N + 2 * (size_t)ny) * sizeof(double)), "hipMalloc");
        hk(hipMalloc((void **)&hb, (size_t)ny * sizeof(double)), "hipMalloc");
        hk(hipMalloc((void **)&ht, (size_t)ny * sizeof(double)), "hipMalloc");
        std::vector<double> id(N, 0.0);
        for (int j = 0; j < ny; ++j) id[(size_t)j * ny + j] = 1.0;       // column ix = j carries the unit vector e_j along y: h(ix, j) = delta
        hk(hipMemcpy(h, id.data(), N * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy");
        hk(hipMemset(hb, 0, (size_t)ny * sizeof(double)), "hipMemset");
        hk(hipMemset(ht, 0, (size_t)ny * sizeof(double)), "hipMemset");
        // the copies above ran on the NULL stream, the routine below runs on the caller's stream (tlab_set_stream), which may be non-blocking: order them
        hk(hipDeviceSynchronize(), "hipDeviceSynchronize");
        ok(tlab_boundary_bcs_neumann_y(d->g[1], ibc, ny, ny, 1, h, hb, ht, tmp), "BOUNDARY_BCS_NEUMANN_Y (weights)");
        hk(hipStreamSynchronize(tlab_current_stream()), "sync");
        std::vector<double> wb(ny), wt(ny);
        hk(hipMemcpy(wb.data(), hb, (size_t)ny * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
        hk(hipMemcpy(wt.data(), ht, (size_t)ny * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
        release();
        h = tmp = hb = ht = nullptr;
        double mb = 0.0, mt = 0.0;
        for (int j = 0; j < ny; ++j) { mb = std::max(mb, std::fabs(wb[j])); mt = std::max(mt, std::fabs(wt[j])); }
        int K = 1;
        for (int j = 0; j < ny; ++j) {
            if ((ibc & 1) && std::fabs(wb[j]) > 1.0e-22 * mb) K = std::max(K, j + 1);
            if ((ibc & 2) && std::fabs(wt[ny - 1 - j]) > 1.0e-22 * mt) K = std::max(K, j + 1);
        }
        for (int j = 0; j < ny; ++j)
            if (!std::isfinite(wb[j]) || !std::isfinite(wt[j])) return false;
        if (((ibc & 1) && mb == 0.0) || ((ibc & 2) && mt == 0.0)) return false;       // weights that are all zero are never cached (the derivative pass serves then)
        // One Neumann wall only: the functional also sees the value on the OPPOSITE (Dirichlet) wall row through the biased closure there, with a weight
        // ~0.38^ny -- and the callers take their sums over tendencies whose Dirichlet wall row is already zeroed.  On lines so short that this weight
        // survives the cut (K reaches the other wall; measured 4.7e-10 at ny = 24) the route is not exact: the derivative pass serves.
        if (ibc != 3 && K >= ny) return false;
        std::vector<double> w((size_t)2 * K, 0.0);
        for (int j = 0; j < K; ++j) { w[j] = (ibc & 1) ? wb[j] : 0.0; w[K + j] = (ibc & 2) ? wt[ny - 1 - j] : 0.0; }
        hk(hipMalloc((void **)&W.w, w.size() * sizeof(double)), "hipMalloc");
        hk(hipMemcpy(W.w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy");
        W.K = K;
        return true;
    } catch (...) {
        release();
        throw;
    }
}
}  // namespace

long long tlab_internal_dns_points(tlab_dns_t d) { return d ? (long long)d->nx * d->ny * d->nz : 0; }      // deferred.cpp
int tlab_internal_dns_nscal(tlab_dns_t d) { return d ? d->nscal : 0; }

extern "C" {

// the wall-plane weights of a Neumann variant for the other drivers of the library (slab.cpp): 1 and (w = [2][K] device weights, K) when available
int tlab_internal_dns_neumann_weights(tlab_dns_t d, int ibc, const double **w, int *K) {
    try {
        if (!d || ibc < 1 || ibc > 3 || !neumann_weights(d, ibc)) return 0;
        *w = d->neuw[ibc].w;
        *K = d->neuw[ibc].K;
        return 1;
    } catch (...) {
        return 0;
    }
}

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
        d->stagger = tlab_fdm_plan_info(gx, 7) == 1 || (nz > 1 && tlab_fdm_plan_info(gz, 7) == 1);
        if (d->stagger && (tlab_fdm_plan_info(gx, 7) != 1 || (nz > 1 && tlab_fdm_plan_info(gz, 7) != 1)))
            throw Fail(TLAB_EINVAL, "staggering: both the x and the z plan need their interpolation tables (tlab_fdm_plan_set_stagger)");
        d->schmidt.assign(schmidt, schmidt + nscal);
        d->scal_jmin.assign(nscal, TLAB_DNS_BCS_DIRICHLET);
        d->scal_jmax.assign(nscal, TLAB_DNS_BCS_DIRICHLET);
        hk(hipMalloc((void **)&d->bcs_hb, (size_t)nx * nz * sizeof(double)), "hipMalloc");
        hk(hipMalloc((void **)&d->bcs_ht, (size_t)nx * nz * sizeof(double)), "hipMalloc");
        // TIME_INITIALIZE, tools/dns/time.f90:138-176
        d->nz_total = gz->t.n;
        double sf = 1.0;                                            // (prandtl = 1: incompressible)
        for (int is = 0; is < nscal; ++is) sf = std::max(sf, 1.0 / schmidt[is]);
        d->schmidtfactor = sf * visc;
        std::vector<double> o2[3];
        tlab_fdm_plan_t gg[3] = {gx, gy, gz};
        for (int ig = 0; ig < 3; ++ig) {
            const int m = gg[ig]->t.n;
            std::vector<double> o1(m);
            o2[ig].resize(m);
            const bool have_jac = (int)gg[ig]->t.jac.size() >= m;      // host-built plans carry it only after tlab_fdm_plan_set_aux
            if (!have_jac) d->dx2i = -1.0;
            for (int i = 0; i < m; ++i) { o1[i] = have_jac ? 1.0 / gg[ig]->t.jac[i] : 0.0; o2[ig][i] = o1[i] * o1[i]; }
            hk(hipMalloc((void **)&d->od[ig], (size_t)m * sizeof(double)), "hipMalloc");
            hk(hipMemcpy(d->od[ig], o1.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy");
        }
        if (d->dx2i >= 0.0) {   // separable maximum of the sum = sum of the maxima over the directions with more than one point
            d->dx2i = 0.0;
            for (int ig = 0; ig < 3; ++ig)
                if (gg[ig]->t.n > 1) d->dx2i += *std::max_element(o2[ig].begin(), o2[ig].end());
        }
        hk(hipMalloc((void **)&d->part, (size_t)2 * 1024 * sizeof(double)), "hipMalloc");
        *out = d.release();
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

int tlab_dns_destroy(tlab_dns_t d) {
    (void)tlab_internal_deferred_flush();
    // the operator state THIS driver switched on goes with it; one set through tlab_opr_burgers_set_anelastic (OPR_Burgers_AMD_Anelastic) or by
    // another driver since is not this driver's to clear
    if (d && d->anel_owner && d->rb) {
        const double *rb, *rib;
        unsigned long ver;
        (void)tlab_internal_anelastic_state(&rb, &rib, &ver);
        if (ver == d->anel_version) (void)tlab_opr_burgers_set_anelastic(0, nullptr, nullptr);
    }
    delete d;
    return TLAB_OK;
}

// The anelastic state lives with the operators (tlab_opr_burgers_set_anelastic = the module variables of OPR_Burgers_Initialize); the driver's density
// weights follow it, whichever entry point set it (tlab_dns_set_anelastic, tlab_opr_burgers_set_anelastic, Fortran OPR_Burgers_AMD_Anelastic), so the
// Burgers operators and the pressure step can never disagree about the equations being solved.
static void follow_anelastic(tlab_dns_t d) {
    const double *rb, *rib;
    unsigned long ver;
    const int ny = tlab_internal_anelastic_state(&rb, &rib, &ver);
    if (ver == d->anel_version) return;
    if (d->rb) { (void)hipFree(d->rb); d->rb = nullptr; }
    if (d->rib) { (void)hipFree(d->rib); d->rib = nullptr; }
    d->rb_wall[0] = d->rb_wall[1] = 1.0;
    if (ny > 0) {
        if (ny != d->ny) throw Fail(TLAB_EINVAL, "the anelastic profiles of the operators (tlab_opr_burgers_set_anelastic) have another ny than this driver");
        const size_t bytes = (size_t)ny * sizeof(double);
        hk(hipMalloc((void **)&d->rb, bytes), "hipMalloc");
        hk(hipMalloc((void **)&d->rib, bytes), "hipMalloc");
        hk(hipMemcpy(d->rb, rb, bytes, hipMemcpyHostToDevice), "hipMemcpy");
        hk(hipMemcpy(d->rib, rib, bytes, hipMemcpyHostToDevice), "hipMemcpy");
        d->rb_wall[0] = rb[0];
        d->rb_wall[1] = rb[ny - 1];
    }
    d->anel_version = ver;
}

// dst += Burgers_dir(s; vel): fused accumulation when the fast kernels apply, otherwise the reference's temp + add path
static void burgers_into(tlab_dns_t d, int dir, double nu, const double *s, const double *vel, bool self, double *dst, double *tmp,
                         double *scratch, bool &pending_add, double **pend, int &npend) {
    const int nx = d->nx, ny = d->ny, nz = d->nz;
    if (d->fuse && !d->rb && !tlab_internal_dealiasing() && tlab_internal_burgers_acc(dir, d->g[dir - 1], nx, ny, nz, 0, nu, s, vel, dst)) return;
    ok(tlab_opr_burgers(dir, d->g[dir - 1], self ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, nx, ny, nz, 0, nu, s, vel, tmp, scratch, 0), "OPR_Burgers");
    pend[npend++] = tmp;
    pending_add = true;
}

static void rhs_impl(tlab_dns_t d, double dte, double *const *q, double *const *s, double *const *hq, double *const *hs,
                     double *const *txc, bool tail_update, double kco, int scale_tendencies) {
    const int nx = d->nx, ny = d->ny, nz = d->nz;
    const long long n = (long long)nx * ny * nz;
    hipStream_t st = tlab_current_stream();
    follow_anelastic(d);
    double *u = q[0], *v = q[1], *w = q[2];
    double *tmp1 = txc[0], *tmp2 = txc[1], *tmp3 = txc[2], *tmp4 = txc[3], *tmp7 = txc[6], *tmp8 = txc[7], *tmp9 = txc[8];
    tlab_fdm_plan_t gx = d->g[0], gy = d->g[1], gz = d->g[2];
    const int B0 = 0;  // bcs = 0: biased, non-zero (:67)
    const double nu = d->visc;
    // ---- diffusion and advection (:98-136), scalars (:149-162).  Reference: every OPR_Burgers result goes to a tmp array and
    // hq = hq + tmp_a + tmp_b + tmp_c afterwards; here each kernel adds its result to hq directly (same summation order)
    // when the fused kernels apply, and falls back to tmp + k_add3 per equation otherwise. ----
    auto surface = [&](int is) { return !d->sfc_jmin.empty() && (d->sfc_jmin[is] == 1 || d->sfc_jmax[is] == 1); };
    bool any_surface = false;
    for (int is = 0; is < d->nscal; ++is) {      // keep the old tendency of the scalar at the boundary for the dynamic BCs (:77-87); zero otherwise
        if (!surface(is)) continue;
        any_surface = true;
        // at the start of a Runge-Kutta step the tendencies COUNT as zero (tlab_dns_begin_step instead of the fill of time.f90:212-216): whatever
        // hs still holds from the last step must not become BcsScal%ref
        if (d->fresh) {
            hk(hipMemsetAsync(d->sref_b[is], 0, (size_t)nx * nz * sizeof(double), st), "memset");
            hk(hipMemsetAsync(d->sref_t[is], 0, (size_t)nx * nz * sizeof(double), st), "memset");
            continue;
        }
        hk(launch_get_wall_planes(hs[is], d->sref_b[is], d->sref_t[is], nx, ny, nz, st), "wall planes");
        if (d->sfc_jmin[is] != 1) hk(hipMemsetAsync(d->sref_b[is], 0, (size_t)nx * nz * sizeof(double), st), "memset");
        if (d->sfc_jmax[is] != 1) hk(hipMemsetAsync(d->sref_t[is], 0, (size_t)nx * nz * sizeof(double), st), "memset");
    }
    struct Eq { double *dst; const double *fld; double nu; int order[3]; };   // order = directions in the reference's summation order
    std::vector<Eq> eqs = {{hq[0], u, nu, {1, 2, 3}},      // hq1 + tmp1(X) + tmp7(Y) + tmp8(Z)   (:110)
                           {hq[1], v, nu, {2, 1, 3}},      // hq2 + tmp2(Y) + tmp7(X) + tmp8(Z)   (:122)
                           {hq[2], w, nu, {3, 1, 2}}};     // hq3 + tmp3(Z) + tmp7(X) + tmp8(Y)   (:134)
    for (int is = 0; is < d->nscal; ++is) eqs.push_back({hs[is], s[is], d->visc / d->schmidt[is], {1, 2, 3}});   // :158, opr_burgers.f90:97
    const double *vel[3] = {u, v, w};
    double *tmps[3] = {tmp1, tmp7, tmp8};
    // Fused path: one launch per direction serves all equations (they share the advecting velocity of that direction), four fields per
    // launch.  The terms of an equation are then added in the order x, y, z instead of the reference's {1,2,3},{2,1,3},{3,1,2}: rounding only.
    // every fused form below assumes the plain operators: anelastic runs and runs with [Dealiasing] take the literal sequence
    const bool anel = d->rb != nullptr;
    const bool stag = d->stagger;                 // staggered pressure grid (rhs_global_incompressible_1.f90:216-226, 266-273, 307-317)
    const bool literal = anel || stag || tlab_internal_dealiasing();
    double *tmp5 = txc[4];
    // (the Burgers launches themselves only care about [Dealiasing]: the anelastic diffusion weight is inside the fused kernels where
    // tlab_internal_burgers_fusable says so, and the staggered pressure grid does not touch them; the epilogues below are incompressible forms)
    auto burgers_ok = [&](int dir, tlab_fdm_plan_t g) {
        return anel ? tlab_internal_burgers_fusable_anelastic(dir, g, nx, ny, nz) : tlab_internal_burgers_fusable(dir, g, nx, ny, nz);
    };
    const bool batched = !tlab_internal_dealiasing() && d->fuse && burgers_ok(1, gx) && burgers_ok(2, gy) && burgers_ok(3, gz);
    const bool fresh = d->fresh;       // TIME_RUNGEKUTTA zeroes hq, hs at the start of a step (time.f90:212-216): the first launch overwrites instead
    d->fresh = false;
    if (fresh && !batched) {
        for (size_t e = 0; e < eqs.size(); ++e) hk(hipMemsetAsync(eqs[e].dst, 0, (size_t)n * sizeof(double), st), "memset");
    }
    // The scalars have no pressure term: with Dirichlet walls their Runge-Kutta update (s += dte hs, hs *= kco; time.f90:645-664, :272-297) can
    // ride on the LAST Burgers launch that adds to hs, if that launch is the x one (the wave-per-line kernel has the registers for it; the
    // y/z tile kernels do not, measured).  The directions then run z, y, x instead of x, y, z: the terms are summed in another order, rounding only.
    static const bool finish_off = [] { const char *e = getenv("TLAB_SCALAR_FINISH"); return e && atoi(e) == 0; }();
    bool finish_scal = !finish_off && batched && !literal && tail_update && d->nscal > 0 && tlab_internal_burgers_can_finish(1, gx, nx, ny, nz);
    // Neumann scalars can ride too where the fused Neumann tail below applies: the epilogue finishes their interior with zero wall tendencies, and the
    // wall planes follow from weighted sums over the stored tendencies next to the walls (neumann_weights; k_wall_fix) instead of a derivative pass
    // over the field (TLAB_NEUMANN_PLANES=0: that pass, k_rtile<P1+neumann final>)
    bool scal_neumann_planes = false;
    {
        auto ibc_of = [](int tmin, int tmax) { return (tmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (tmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0); };
        bool any_neu = false, scal_neu = false;
        for (int iq = 0; iq < 3; ++iq) any_neu = any_neu || ibc_of(d->flow_jmin[iq], d->flow_jmax[iq]) != 0;
        for (int is = 0; is < d->nscal; ++is) scal_neu = scal_neu || ibc_of(d->scal_jmin[is], d->scal_jmax[is]) != 0;
        any_neu = any_neu || scal_neu;
        const char *npe = getenv("TLAB_NEUMANN_PLANES");
        const bool tail_fast = any_neu && tail_update && d->fuse && !literal && nz > 1 && !d->pfilter[0] && !d->pfilter[1] && !d->pfilter[2] && !any_surface &&
                               tlab_internal_poisson_can_v_final(d->poisson) && tlab_internal_neumann_final_ok(gy, nx, ny, nz) &&
                               tlab_internal_partial_p1_fusable(1, gx, nx, ny, nz) && tlab_internal_partial_p1_fusable(3, gz, nx, ny, nz);      // = neu_fast below
        if (finish_scal && scal_neu && tail_fast && !(npe && atoi(npe) == 0)) {
            scal_neumann_planes = true;
            for (int is = 0; is < d->nscal; ++is) {
                const int ibc = ibc_of(d->scal_jmin[is], d->scal_jmax[is]);
                if (ibc != 0 && !neumann_weights(d, ibc)) scal_neumann_planes = false;
            }
        }
    }
    for (int is = 0; is < d->nscal && !scal_neumann_planes; ++is)
        finish_scal = finish_scal && d->scal_jmin[is] == TLAB_DNS_BCS_DIRICHLET && d->scal_jmax[is] == TLAB_DNS_BCS_DIRICHLET;
    finish_scal = finish_scal && !any_surface;      // the wall planes of a scalar with a surface model are not zero
    // Likewise the x term of the pressure forcing, d/dx (hq1 + u/dte) (:197-230): when the x Burgers launch runs last it holds the finished
    // tendency of u in registers, line by line, and differentiates it on the spot instead of a separate launch re-reading hq1 and u.
    const double idte = d->remove_divergence ? 1.0 / dte : 0.0;      // hq + 0 q is hq bit for bit: the same kernels serve the else-branch (:234-250)
    const bool x_last = !finish_off && batched && !literal && tlab_internal_burgers_can_finish(1, gx, nx, ny, nz);
    const bool div_in_burgers = x_last && d->fuse && tlab_internal_partial_p1_fusable(2, gy, nx, ny, nz) && tlab_internal_partial_p1_fusable(3, gz, nx, ny, nz);
    // ... and the y and z terms, d/dy (hq2 + v/dte) and d/dz (hq3 + w/dte): each component gets the term of its OWN direction last, in a launch of
    // its own whose workgroups hold the finished tendency line by line and differentiate it on the spot (k_htile<BURGERS+div>).  Order of the
    // launches: z (all but w), y (all but v), x (all: finishes u, the scalars, writes the x term), y (v: adds the y term), z (w: adds the z term).
    // Against the three four-field launches + two forcing passes this saves the re-read of hq2, v, hq3, w (four passes of the eight).
    const bool div_all = div_in_burgers && nz > 1 && tlab_internal_burgers_can_div(2, gy, nx, ny, nz) && tlab_internal_burgers_can_div(3, gz, nx, ny, nz);
    if (batched && div_all) {
        struct Launch { int dir; std::vector<int> eq; bool last_x; int divf; };      // divf: equation whose forcing term rides on the launch (-1: none)
        std::vector<int> all_but_w, all_but_v, all;
        for (int e = 0; e < (int)eqs.size(); ++e) { all.push_back(e); if (e != 2) all_but_w.push_back(e); if (e != 1) all_but_v.push_back(e); }
        const Launch plan[5] = {{3, all_but_w, false, -1}, {2, all_but_v, false, -1}, {1, all, true, 0}, {2, {1}, false, 1}, {3, {2}, false, 2}};
        bool touched[16] = {false};
        for (const Launch &L : plan) {
            for (size_t e0 = 0; e0 < L.eq.size(); e0 += 4) {
                const int nf = (int)std::min<size_t>(4, L.eq.size() - e0);
                const double *sp[4]; double *rp[4]; double nup[4];
                int fin[4] = {0, 0, 0, 0};
                unsigned fresh_mask = 0;
                bool all_fresh = fresh, any_fresh = false;
                for (int f = 0; f < nf; ++f) {
                    const int e = L.eq[e0 + f];
                    sp[f] = eqs[e].fld; rp[f] = eqs[e].dst; nup[f] = eqs[e].nu;
                    fin[f] = (finish_scal && L.last_x && e >= 3) ? 1 : 0;        // equations 3.. are the scalars
                    const bool fr = fresh && !touched[e];
                    if (fr) { fresh_mask |= 1u << f; any_fresh = true; } else all_fresh = false;
                    touched[e] = true;
                }
                const bool any_fin = fin[0] || fin[1] || fin[2] || fin[3];
                const bool over = all_fresh && any_fresh;                   // every field of the launch starts its tendency here
                double *divp = nullptr;
                if (L.divf >= 0 && (L.dir != 1 || e0 == 0)) divp = tmp1;  // x: batch 0 holds u; y, z: the one-field launches
                if (!tlab_internal_burgers_acc_n(L.dir, d->g[L.dir - 1], nx, ny, nz, 0, nf, nup, sp, vel[L.dir - 1], rp, over,
                                                 any_fin ? fin : nullptr, dte, kco, scale_tendencies ? 1 : 0, divp, idte, over ? 0u : fresh_mask))
                    throw Fail(TLAB_EINVAL, "internal: inconsistent fused Burgers path");
            }
        }
    } else if (batched) {
        const int order_xyz[3] = {1, 2, 3}, order_zyx[3] = {3, 2, 1};
        const int *order = x_last ? order_zyx : order_xyz;
        for (int k = 0; k < 3; ++k) {
            const int dir = order[k];
            for (size_t e0 = 0; e0 < eqs.size(); e0 += 4) {
                const int nf = (int)std::min<size_t>(4, eqs.size() - e0);
                const double *sp[4]; double *rp[4]; double nup[4];
                int fin[4] = {0, 0, 0, 0};
                for (int f = 0; f < nf; ++f) {
                    sp[f] = eqs[e0 + f].fld; rp[f] = eqs[e0 + f].dst; nup[f] = eqs[e0 + f].nu;
                    fin[f] = (finish_scal && k == 2 && e0 + f >= 3) ? 1 : 0;        // equations 3.. are the scalars
                }
                const bool any_fin = fin[0] || fin[1] || fin[2] || fin[3];
                double *divx = (div_in_burgers && dir == 1 && e0 == 0) ? tmp1 : nullptr;       // batch 0 holds u
                if (!tlab_internal_burgers_acc_n(dir, d->g[dir - 1], nx, ny, nz, 0, nf, nup, sp, vel[dir - 1], rp, fresh && k == 0,
                                                 any_fin ? fin : nullptr, dte, kco, scale_tendencies ? 1 : 0, divx, idte, 0u, anel ? d->rib : nullptr))
                    throw Fail(TLAB_EINVAL, "internal: inconsistent fused Burgers path");
            }
        }
    }
    for (size_t e = 0; e < eqs.size() && !batched; ++e) {
        bool pending = false;
        double *pend[3];
        int npend = 0;
        for (int k = 0; k < 3; ++k) {
            const int dir = eqs[e].order[k];
            burgers_into(d, dir, eqs[e].nu, eqs[e].fld, vel[dir - 1], eqs[e].fld == vel[dir - 1], eqs[e].dst, tmps[k], tmp9, pending, pend, npend);
        }
        if (pending) {
            if (npend == 3) hk(launch_add3(eqs[e].dst, pend[0], pend[1], pend[2], n, st), "add3");
            else {   // mixed: add the temporaries one at a time (k_add3 with zero-sized partners is not worth a kernel)
                hk(hipMemsetAsync(tmp9, 0, (size_t)n * sizeof(double), st), "memset");
                hk(launch_add3(eqs[e].dst, pend[0], npend > 1 ? pend[1] : tmp9, tmp9, n, st), "add3");
            }
        }
    }
    // ---- pressure (:177-260, remove_divergence branch): forcing = div(hq + q/dte) ----
    bool fused_div = !literal && d->fuse && tlab_internal_partial_p1_fusable(1, gx, nx, ny, nz) && tlab_internal_partial_p1_fusable(2, gy, nx, ny, nz) &&
                     tlab_internal_partial_p1_fusable(3, gz, nx, ny, nz);
    if (batched && div_all) {
        fused_div = true;     // tmp1 holds the three terms
    } else if (div_in_burgers) {     // tmp1 holds the x term already
        const bool oky = tlab_internal_partial_p1_fused(2, gy, nx, ny, nz, B0, hq[1], v, idte, tmp1, true);
        const bool okz = oky && tlab_internal_partial_p1_fused(3, gz, nx, ny, nz, B0, hq[2], w, idte, tmp1, true);
        if (!oky || !okz) throw Fail(TLAB_EINVAL, "internal: inconsistent fused divergence path");
        fused_div = true;
    } else if (fused_div) {
        fused_div = tlab_internal_partial_p1_fused(2, gy, nx, ny, nz, B0, hq[1], v, idte, tmp1, false);
        if (fused_div) {
            const bool okx = tlab_internal_partial_p1_fused(1, gx, nx, ny, nz, B0, hq[0], u, idte, tmp1, true);
            const bool okz = okx && tlab_internal_partial_p1_fused(3, gz, nx, ny, nz, B0, hq[2], w, idte, tmp1, true);
            if (!okx || !okz) throw Fail(TLAB_EINVAL, "internal: inconsistent fused divergence path");
        }
    }
    if (!fused_div) {
        if (anel && d->fuse) {      // ... with Thermo_Anelastic_WEIGHT_INPLACE(.., rbackground, tmp2 | tmp3 | tmp4) (:211-214) in the same pass
            hk(launch_axpy3w(tmp2, tmp3, tmp4, hq[1], hq[0], hq[2], v, u, w, idte, d->rb, nx, ny, n, st), "axpy3w");
        } else {
            hk(launch_axpy3(tmp2, tmp3, tmp4, hq[1], hq[0], hq[2], v, u, w, idte, n, st), "axpy3");
            if (anel) {      // Thermo_Anelastic_WEIGHT_INPLACE(.., rbackground, tmp2 | tmp3 | tmp4)  (:211-214)
                hk(launch_weight_y(tmp2, tmp2, d->rb, nx, ny, n, 0, st), "weight");
                hk(launch_weight_y(tmp3, tmp3, d->rb, nx, ny, n, 0, st), "weight");
                hk(launch_weight_y(tmp4, tmp4, d->rb, nx, ny, n, 0, st), "weight");
            }
        }
        if (stag) {      // the three terms on the horizontal pressure nodes (:216-226)
            ok(tlab_opr_partial(1, gx, TLAB_OPR_P0_INT_VP, nx, ny, nz, B0, tmp2, tmp5, nullptr), "OPR_Partial_X(P0_INT_VP)");      // Oy derivative
            ok(tlab_opr_partial(2, gy, TLAB_OPR_P1, nx, ny, nz, B0, tmp5, tmp2, nullptr), "OPR_Partial_Y");
            ok(tlab_opr_partial(3, gz, TLAB_OPR_P0_INT_VP, nx, ny, nz, B0, tmp2, tmp1, nullptr), "OPR_Partial_Z(P0_INT_VP)");
            ok(tlab_opr_partial(1, gx, TLAB_OPR_P1_INT_VP, nx, ny, nz, B0, tmp3, tmp5, nullptr), "OPR_Partial_X(P1_INT_VP)");      // Ox derivative
            ok(tlab_opr_partial(3, gz, TLAB_OPR_P0_INT_VP, nx, ny, nz, B0, tmp5, tmp2, nullptr), "OPR_Partial_Z(P0_INT_VP)");
            ok(tlab_opr_partial(1, gx, TLAB_OPR_P0_INT_VP, nx, ny, nz, B0, tmp4, tmp5, nullptr), "OPR_Partial_X(P0_INT_VP)");      // Oz derivative
            ok(tlab_opr_partial(3, gz, TLAB_OPR_P1_INT_VP, nx, ny, nz, B0, tmp5, tmp3, nullptr), "OPR_Partial_Z(P1_INT_VP)");
        } else {
            ok(tlab_opr_partial(2, gy, TLAB_OPR_P1, nx, ny, nz, B0, tmp2, tmp1, nullptr), "OPR_Partial_Y");
            ok(tlab_opr_partial(1, gx, TLAB_OPR_P1, nx, ny, nz, B0, tmp3, tmp2, nullptr), "OPR_Partial_X");
            ok(tlab_opr_partial(3, gz, TLAB_OPR_P1, nx, ny, nz, B0, tmp4, tmp3, nullptr), "OPR_Partial_Z");
        }
        hk(launch_sum3(tmp1, tmp2, tmp3, n, st), "sum3");
    }
    // Neumann BCs in d/dy(p) s.t. v = 0 (:263-281); staggered: the planes of hq2 interpolated onto the pressure nodes (:266-269)
    if (stag) {
        ok(tlab_opr_partial(1, gx, TLAB_OPR_P0_INT_VP, nx, ny, nz, B0, hq[1], tmp5, nullptr), "OPR_Partial_X(P0_INT_VP)");
        ok(tlab_opr_partial(3, gz, TLAB_OPR_P0_INT_VP, nx, ny, nz, B0, tmp5, tmp4, nullptr), "OPR_Partial_Z(P0_INT_VP)");
    }
    hk(launch_get_wall_planes(stag ? tmp4 : hq[1], d->bcs_hb, d->bcs_ht, nx, ny, nz, st), "wall planes");
    if (anel) {          // BcsFlowJmin%ref(:,:,2) = p_bcs(:,1,:) * rbackground(1), Jmax likewise (:275-277)
        hk(launch_scale(d->bcs_hb, d->rb_wall[0], (long long)nx * nz, st), "scale");
        hk(launch_scale(d->bcs_ht, d->rb_wall[1], (long long)nx * nz, st), "scale");
    }
    // pressure in tmp1, Oy derivative in tmp3 (:284).  With Dirichlet walls for v and the RK update folded in, the last inverse transform of the
    // solver finishes the v equation itself (hq2 -= dp/dy; wall planes; v += dte hq2; hq2 *= kco) and tmp3 is not written
    bool v_final = false;
    bool walls_dirichlet = true;      // (a Neumann wall of any component sends all three through the unfused subtraction below)
    for (int iq = 0; iq < 3; ++iq) walls_dirichlet = walls_dirichlet && d->flow_jmin[iq] == TLAB_DNS_BCS_DIRICHLET && d->flow_jmax[iq] == TLAB_DNS_BCS_DIRICHLET;
    // Neumann walls somewhere (free-slip u, w; Neumann scalars): the tail below then runs per field -- gradient subtracted in its own launch,
    // BOUNDARY_BCS_NEUMANN_Y + final update in ONE launch along y (tlab_internal_neumann_final) -- and v, always Dirichlet, is finished by the solver
    auto ibc_y = [](int tmin, int tmax) { return (tmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (tmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0); };
    bool any_neumann = false;
    for (int iq = 0; iq < 3; ++iq) any_neumann = any_neumann || ibc_y(d->flow_jmin[iq], d->flow_jmax[iq]) != 0;
    for (int is = 0; is < d->nscal; ++is) any_neumann = any_neumann || ibc_y(d->scal_jmin[is], d->scal_jmax[is]) != 0;
    const bool neu_fast = any_neumann && tail_update && d->fuse && !literal && nz > 1 && !d->pfilter[0] && !d->pfilter[1] && !d->pfilter[2] && !any_surface &&
                          tlab_internal_poisson_can_v_final(d->poisson) && tlab_internal_neumann_final_ok(gy, nx, ny, nz) &&
                          tlab_internal_partial_p1_fusable(1, gx, nx, ny, nz) && tlab_internal_partial_p1_fusable(3, gz, nx, ny, nz);
    if (tail_update && d->fuse && !literal && !d->pfilter[0] && !d->pfilter[1] && !d->pfilter[2] && (walls_dirichlet || neu_fast) &&
        tlab_internal_poisson_can_v_final(d->poisson)) {
        tlab_internal_poisson_arm_v_final(d->poisson, q[1], hq[1], dte, kco, scale_tendencies);
        v_final = true;
    }
    ok(tlab_opr_poisson(d->poisson, nx, ny, nz, TLAB_BCS_NN, tmp1, tmp2, tmp4, d->bcs_hb, d->bcs_ht, tmp3), "OPR_Poisson");
    if (d->pfilter[0] || d->pfilter[1] || d->pfilter[2]) {      // filter pressure p and its vertical gradient dpdy (:286-290)
        ok(tlab_opr_filter(nx, ny, nz, d->pfilter[0], d->pfilter[1], d->pfilter[2], d->pfilter_rep, tmp1, tmp4), "OPR_FILTER(p)");
        ok(tlab_opr_filter(nx, ny, nz, d->pfilter[0], d->pfilter[1], d->pfilter[2], d->pfilter_rep, tmp3, tmp4), "OPR_FILTER(dpdy)");
    }
    if (scal_neumann_planes && !neu_fast) throw Fail(TLAB_EINVAL, "internal: Neumann scalars were finished without their wall planes");
    if (neu_fast) {
        const char *npe = getenv("TLAB_NEUMANN_PLANES");      // (read per substep: A/B in one process)
        const bool neu_planes = !(npe && atoi(npe) == 0);
        for (int iq = 0; iq < 3; iq += 2) {      // u along x, w along z
            const int dir = iq == 0 ? 1 : 3;
            tlab_fdm_plan_t gd = iq == 0 ? gx : gz;
            const int ibc = ibc_y(d->flow_jmin[iq], d->flow_jmax[iq]);
            bool done;
            if (ibc == 0) done = tlab_internal_gradient_final(dir, gd, nx, ny, nz, tmp1, q[iq], hq[iq], dte, kco, scale_tendencies);
            else if (neu_planes && neumann_weights(d, ibc)) {
                // wall tendencies from weighted sums over the rows next to the walls (of hq and of p; the derivative of the p planes commutes
                // with the sum), then the gradient kernel finishes the field as it does with Dirichlet walls -- one pass over hq instead of two
                const tlab_dns::NeuW &W = d->neuw[ibc];
                const size_t np = (size_t)nx * nz;
                if (!d->wall_planes) hk(hipMalloc((void **)&d->wall_planes, 6 * np * sizeof(double)), "hipMalloc");
                double *Hb = d->wall_planes, *Ht = Hb + np, *Pb = Ht + np, *Pt = Pb + np, *Db = Pt + np, *Dt = Db + np;
                hk(launch_wall_weighted(hq[iq], tmp1, W.w, W.w + W.K, W.K, Hb, Ht, Pb, Pt, nx, ny, nz, st), "wall planes");
                ok(tlab_opr_partial(dir, gd, TLAB_OPR_P1, nx, 1, nz, 0, Pb, Db, nullptr), "OPR_Partial (wall plane)");
                ok(tlab_opr_partial(dir, gd, TLAB_OPR_P1, nx, 1, nz, 0, Pt, Dt, nullptr), "OPR_Partial (wall plane)");
                hk(launch_sub2(Hb, Hb, Db, (long long)np, st), "wall planes");
                hk(launch_sub2(Ht, Ht, Dt, (long long)np, st), "wall planes");
                done = tlab_internal_gradient_final(dir, gd, nx, ny, nz, tmp1, q[iq], hq[iq], dte, kco, scale_tendencies, (ibc & 1) ? Hb : nullptr,
                                                    (ibc & 2) ? Ht : nullptr);
            } else done = tlab_internal_partial_p1_sub(dir, gd, nx, ny, nz, tmp1, hq[iq]) &&
                          tlab_internal_neumann_final(gy, nx, ny, nz, ibc, hq[iq], q[iq], dte, kco, scale_tendencies);
            if (!done) throw Fail(TLAB_EINVAL, "internal: inconsistent fused Neumann tail");
        }
        for (int is = 0; is < d->nscal && !finish_scal; ++is) {
            const int ibc = ibc_y(d->scal_jmin[is], d->scal_jmax[is]);
            if (ibc == 0) hk(launch_final_update(s[is], hs[is], nullptr, nullptr, nullptr, dte, kco, scale_tendencies, nx, ny, nz, st), "final update");
            else if (!tlab_internal_neumann_final(gy, nx, ny, nz, ibc, hs[is], s[is], dte, kco, scale_tendencies))
                throw Fail(TLAB_EINVAL, "internal: inconsistent fused Neumann tail");
        }
        for (int is = 0; is < d->nscal && finish_scal && scal_neumann_planes; ++is) {      // finished by the x Burgers launch but for their wall planes
            const int ibc = ibc_y(d->scal_jmin[is], d->scal_jmax[is]);
            if (ibc == 0) continue;
            const tlab_dns::NeuW &W = d->neuw[ibc];
            const size_t np = (size_t)nx * nz;
            if (!d->wall_planes) hk(hipMalloc((void **)&d->wall_planes, 6 * np * sizeof(double)), "hipMalloc");
            double *Sb = d->wall_planes, *St = Sb + np;
            hk(launch_wall_weighted(hs[is], nullptr, W.w, W.w + W.K, W.K, Sb, St, nullptr, nullptr, nx, ny, nz, st), "wall planes");
            hk(launch_wall_fix(s[is], hs[is], (ibc & 1) ? Sb : nullptr, (ibc & 2) ? St : nullptr, dte, kco, scale_tendencies, nx, ny, nz, st), "wall planes");
        }
        return;
    }
    // ---- pressure gradient (:319-320).  With Dirichlet walls and the RK update folded in (tail_update), the x- and z-gradient kernels
    // finish u and w themselves: hq -= dp/dx; wall planes; q += dte hq; hq *= kco (no gradient array is written or re-read) ----
    bool grad_final = false, grad_sub = false;
    if (tail_update && d->fuse && nz > 1 && !literal) {
        bool dirichlet = true;
        for (int iq = 0; iq < 3; ++iq) dirichlet = dirichlet && d->flow_jmin[iq] == TLAB_DNS_BCS_DIRICHLET && d->flow_jmax[iq] == TLAB_DNS_BCS_DIRICHLET;
        if (dirichlet && tlab_internal_partial_p1_fusable(1, gx, nx, ny, nz) && tlab_internal_partial_p1_fusable(3, gz, nx, ny, nz)) {
            const bool okx = tlab_internal_gradient_final(1, gx, nx, ny, nz, tmp1, q[0], hq[0], dte, kco, scale_tendencies);
            const bool okz = okx && tlab_internal_gradient_final(3, gz, nx, ny, nz, tmp1, q[2], hq[2], dte, kco, scale_tendencies);
            if (okx != okz) throw Fail(TLAB_EINVAL, "internal: inconsistent fused gradient path");
            grad_final = okx;
        }
    }
    if (stag) {          // back onto the horizontal velocity nodes (:307-317)
        ok(tlab_opr_partial(3, gz, TLAB_OPR_P0_INT_PV, nx, ny, nz, B0, tmp3, tmp5, nullptr), "OPR_Partial_Z(P0_INT_PV)");       // dp/dy
        ok(tlab_opr_partial(1, gx, TLAB_OPR_P0_INT_PV, nx, ny, nz, B0, tmp5, tmp3, nullptr), "OPR_Partial_X(P0_INT_PV)");
        ok(tlab_opr_partial(3, gz, TLAB_OPR_P1_INT_PV, nx, ny, nz, B0, tmp1, tmp5, nullptr), "OPR_Partial_Z(P1_INT_PV)");       // dp/dz
        ok(tlab_opr_partial(1, gx, TLAB_OPR_P0_INT_PV, nx, ny, nz, B0, tmp5, tmp4, nullptr), "OPR_Partial_X(P0_INT_PV)");
        ok(tlab_opr_partial(3, gz, TLAB_OPR_P0_INT_PV, nx, ny, nz, B0, tmp1, tmp5, nullptr), "OPR_Partial_Z(P0_INT_PV)");       // dp/dx
        ok(tlab_opr_partial(1, gx, TLAB_OPR_P1_INT_PV, nx, ny, nz, B0, tmp5, tmp2, nullptr), "OPR_Partial_X(P1_INT_PV)");
    } else if (!grad_final) {
        // the gradient launches subtract themselves from the tendencies where the fused kernels apply (hq -= dp/dx: the same difference k_sub3 forms
        // from a stored gradient, two passes per component less); this is the tail of the RHS-only entry, i.e. of an unpatched Fortran host
        if (d->fuse && !literal && nz > 1 && tlab_internal_partial_p1_fusable(1, gx, nx, ny, nz) && tlab_internal_partial_p1_fusable(3, gz, nx, ny, nz)) {
            const bool okx = tlab_internal_partial_p1_sub(1, gx, nx, ny, nz, tmp1, hq[0]);
            const bool okz = okx && tlab_internal_partial_p1_sub(3, gz, nx, ny, nz, tmp1, hq[2]);
            if (!okx || !okz) throw Fail(TLAB_EINVAL, "internal: inconsistent fused gradient path");
            if (!v_final) hk(launch_axpy1(hq[1], hq[1], tmp3, -1.0, n, st), "axpy1");      // hq2 - dp/dy
            grad_sub = true;
        } else {
            ok(tlab_opr_partial(1, gx, TLAB_OPR_P1, nx, ny, nz, B0, tmp1, tmp2, nullptr), "OPR_Partial_X(p)");
            ok(tlab_opr_partial(3, gz, TLAB_OPR_P1, nx, ny, nz, B0, tmp1, tmp4, nullptr), "OPR_Partial_Z(p)");
        }
    }
    // ---- boundary conditions (:360-398): Dirichlet -> the tendency vanishes on the wall plane; Neumann -> the wall tendency
    // keeps d/dy = 0 there (BOUNDARY_BCS_NEUMANN_Y on the finished tendency; tmp1 is its work array as in the reference) ----
    auto ibc_of = [](int tmin, int tmax) { return (tmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (tmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0); };
    int ibc_q[3], any_q = 0;
    for (int iq = 0; iq < 3; ++iq) any_q |= (ibc_q[iq] = ibc_of(d->flow_jmin[iq], d->flow_jmax[iq]));
    // planes of a field: Neumann values where selected, zeros (null) elsewhere
    auto planes = [&](int ibc, const double *h, const double *&pb, const double *&pt) {
        pb = pt = nullptr;
        if (ibc == 0) return;
        ok(tlab_boundary_bcs_neumann_y(gy, ibc, nx, ny, nz, h, d->bcs_hb, d->bcs_ht, tmp1), "BOUNDARY_BCS_NEUMANN_Y");
        if (ibc & 1) pb = d->bcs_hb;
        if (ibc & 2) pt = d->bcs_ht;
    };
    const double *pb, *pt;
    // planes of a scalar: Neumann values where selected, then the dynamic surface model (BOUNDARY_BCS_SURFACE_Y, :393-396), zeros (null) elsewhere
    auto scal_planes = [&](int is, const double *&qb, const double *&qt) {
        const int ibc = ibc_of(d->scal_jmin[is], d->scal_jmax[is]);
        planes(ibc, hs[is], qb, qt);
        if (!surface(is)) return;
        const size_t pbytes = (size_t)nx * nz * sizeof(double);
        if (qb) hk(hipMemcpyAsync(d->sref_b[is], qb, pbytes, hipMemcpyDeviceToDevice, st), "memcpy");      // the Neumann value replaces the kept tendency
        if (qt) hk(hipMemcpyAsync(d->sref_t[is], qt, pbytes, hipMemcpyDeviceToDevice, st), "memcpy");
        const double diff = d->visc / d->schmidt[is];
        ok(tlab_opr_partial(2, gy, TLAB_OPR_P1, nx, ny, nz, B0, s[is], tmp1, nullptr), "OPR_Partial_Y (surface flux)");      // boundary_bcs.f90:508
        // AVG1V2D(imax, jmax, kmax, 1, 1, tmp1) at BOTH ends (:520, :535): the plane j = 1
        if (d->sfc_jmin[is] == 1) hk(launch_surface_flux(d->sref_b[is], tmp1, 0, 0, 1.0, diff, d->cpl_jmin[is], d->sfc_avg, nx, ny, nz, st), "surface flux");
        if (d->sfc_jmax[is] == 1) hk(launch_surface_flux(d->sref_t[is], tmp1, ny - 1, 0, -1.0, diff, d->cpl_jmax[is], d->sfc_avg, nx, ny, nz, st), "surface flux");
        qb = d->sref_b[is]; qt = d->sref_t[is];
    };
    // Thermo_Anelastic_WEIGHT_SUBTRACT(.., ribackground, tmp2 | tmp3 | tmp4, hq(:,1) | hq(:,2) | hq(:,3))  (:326-329): a pass of its own, or -- for a
    // component with Dirichlet walls whose Runge-Kutta update follows -- the operand of that update (k_final_update with gw)
    bool gw_defer[3] = {false, false, false};
    if (anel) {
        double *gt[3] = {tmp2, tmp3, tmp4};
        for (int iq = 0; iq < 3; ++iq) {
            gw_defer[iq] = tail_update && d->fuse && ibc_q[iq] == 0;
            if (!gw_defer[iq]) hk(launch_weight_y(hq[iq], gt[iq], d->rib, nx, ny, n, 1, st), "weight");
        }
    }
    if (tail_update) {
        // hq -= grad p (:348-352), wall planes (:373-375), q += dte hq (time.f90:645-664), hq *= kco (:272-297) in one pass per field
        double *gp[3] = {tmp2, tmp3, tmp4};
        if (anel || grad_sub) gp[0] = gp[1] = gp[2] = nullptr;      // subtracted above (anelastic: with the density weight)
        if (any_q && !anel && !grad_sub) {   // the Neumann planes need the finished tendency first
            hk(launch_sub3(hq[0], hq[1], hq[2], tmp2, tmp3, tmp4, n, st), "sub3");
            gp[0] = gp[1] = gp[2] = nullptr;
        }
        for (int iq = 0; iq < 3; ++iq) {
            if (grad_final && iq != 1) continue;          // u and w are finished already
            if (v_final && iq == 1) continue;             // v too (inside OPR_Poisson)
            planes(ibc_q[iq], hq[iq], pb, pt);
            if (gw_defer[iq]) {
                double *gt[3] = {tmp2, tmp3, tmp4};
                hk(launch_final_update(q[iq], hq[iq], gt[iq], pb, pt, dte, kco, scale_tendencies, nx, ny, nz, st, d->rib), "final update");
            } else {
                hk(launch_final_update(q[iq], hq[iq], gp[iq], pb, pt, dte, kco, scale_tendencies, nx, ny, nz, st), "final update");
            }
        }
        for (int is = 0; is < d->nscal && !finish_scal; ++is) {      // (finish_scal: done in the epilogue of the x Burgers launch)
            scal_planes(is, pb, pt);
            hk(launch_final_update(s[is], hs[is], nullptr, pb, pt, dte, kco, scale_tendencies, nx, ny, nz, st), "final update");
        }
    } else {
        if (!anel && !grad_sub) hk(launch_sub3(hq[0], hq[1], hq[2], tmp2, tmp3, tmp4, n, st), "sub3");
        for (int iq = 0; iq < 3; ++iq) {
            planes(ibc_q[iq], hq[iq], pb, pt);
            hk(launch_set_wall_planes(hq[iq], pb, pt, nx, ny, nz, st), "wall planes");
        }
        for (int is = 0; is < d->nscal; ++is) {
            scal_planes(is, pb, pt);
            hk(launch_set_wall_planes(hs[is], pb, pt, nx, ny, nz, st), "wall planes");
        }
    }
}

int tlab_rhs_global_incompressible_1(tlab_dns_t d, double dte, double *const *q, double *const *s, double *const *hq,
                                     double *const *hs, double *const *txc) {
    try {
        if (!d || !q || !hq || !txc || (d->nscal > 0 && (!s || !hs)) || dte <= 0.0) throw Fail(TLAB_EINVAL, "tlab_rhs_global_incompressible_1: bad arguments");
        rhs_impl(d, dte, q, s, hq, hs, txc, false, 1.0, 0);
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

int tlab_time_substep_incompressible_explicit(tlab_dns_t d, double dte, double kco, int scale_tendencies, double *const *q,
                                              double *const *s, double *const *hq, double *const *hs, double *const *txc) {
    try {
        if (!d || !q || !hq || !txc || (d->nscal > 0 && (!s || !hs)) || dte <= 0.0) throw Fail(TLAB_EINVAL, "tlab_time_substep_incompressible_explicit: bad arguments");
        rhs_impl(d, dte, q, s, hq, hs, txc, true, kco, scale_tendencies);
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

int tlab_dns_set_bcs(tlab_dns_t d, const int *flow_jmin, const int *flow_jmax, const int *scal_jmin, const int *scal_jmax) {
    (void)tlab_internal_deferred_flush();
    auto valid = [](int t) { return t == TLAB_DNS_BCS_DIRICHLET || t == TLAB_DNS_BCS_NEUMANN; };
    if (!d || !flow_jmin || !flow_jmax || (d->nscal > 0 && (!scal_jmin || !scal_jmax))) {
        tlab_set_error("tlab_dns_set_bcs: bad arguments");
        return TLAB_EINVAL;
    }
    for (int i = 0; i < 3; ++i)
        if (!valid(flow_jmin[i]) || !valid(flow_jmax[i])) { tlab_set_error("tlab_dns_set_bcs: type must be DNS_BCS_DIRICHLET or DNS_BCS_NEUMANN"); return TLAB_EINVAL; }
    for (int i = 0; i < d->nscal; ++i)
        if (!valid(scal_jmin[i]) || !valid(scal_jmax[i])) { tlab_set_error("tlab_dns_set_bcs: type must be DNS_BCS_DIRICHLET or DNS_BCS_NEUMANN"); return TLAB_EINVAL; }
    if (flow_jmin[1] != TLAB_DNS_BCS_DIRICHLET || flow_jmax[1] != TLAB_DNS_BCS_DIRICHLET) {
        tlab_set_error("tlab_dns_set_bcs: the wall-normal velocity must be Dirichlet (impermeable walls; the pressure BCs assume v = 0)");
        return TLAB_EUNSUPPORTED;
    }
    for (int i = 0; i < 3; ++i) { d->flow_jmin[i] = flow_jmin[i]; d->flow_jmax[i] = flow_jmax[i]; }
    for (int i = 0; i < d->nscal; ++i) { d->scal_jmin[i] = scal_jmin[i]; d->scal_jmax[i] = scal_jmax[i]; }
    // the wall-plane weights of the Neumann variants in use are built HERE (allocations, a synchronisation), not in the middle of the first substep
    try {
        auto variant = [](int jmin, int jmax) { return (jmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (jmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0); };
        for (int i = 0; i < 3; ++i)
            if (variant(flow_jmin[i], flow_jmax[i])) (void)neumann_weights(d, variant(flow_jmin[i], flow_jmax[i]));
        for (int i = 0; i < d->nscal; ++i)
            if (variant(scal_jmin[i], scal_jmax[i])) (void)neumann_weights(d, variant(scal_jmin[i], scal_jmax[i]));
    } catch (const Fail &e) {
        tlab_set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EHIP;
    }
    return TLAB_OK;
}

// min / max of a device array through per-block partials reduced on the host (diagnostics: once per time step, not per substep)
static void minmax_impl(tlab_dns_t d, const double *a, const double *v, const double *w, int mode, int nx, int ny, int nz, double *mn, double *mx) {
    hipStream_t st = tlab_current_stream();
    const long long n = (long long)nx * ny * nz;
    const int nb = (int)std::min<long long>(1024, (n + 255) / 256);
    hk(launch_minmax_partial(a, v, w, d->od[0], d->od[1], d->od[2], mode, nx, ny, nz, d->koffset, d->nz_total > 1 ? 1 : 0, d->part, nb, st), "k_minmax_partial");
    std::vector<double> h((size_t)2 * nb);
    hk(hipMemcpyAsync(h.data(), d->part, (size_t)2 * nb * sizeof(double), hipMemcpyDeviceToHost, st), "hipMemcpy");
    hk(hipStreamSynchronize(st), "sync");
    *mn = *std::min_element(h.begin(), h.begin() + nb);
    *mx = *std::max_element(h.begin() + nb, h.end());
}

// BcsScalJmin/Jmax%SfcType and %cpl of the scalars ([BoundaryConditions] Scalar<i>SfcTypeJmin/Jmax = static | linear, Scalar<i>CouplingJmin/Jmax;
// boundary_bcs.f90:76-87): 0 = DNS_SFC_STATIC, 1 = DNS_SFC_LINEAR.  Single-domain driver only (the plane average is an all-reduce in a decomposed run).
int tlab_dns_set_surface_bcs(tlab_dns_t d, const int *sfc_jmin, const int *sfc_jmax, const double *cpl_jmin, const double *cpl_jmax) {
    (void)tlab_internal_deferred_flush();
    if (!d || (d->nscal > 0 && (!sfc_jmin || !sfc_jmax || !cpl_jmin || !cpl_jmax))) {
        tlab_set_error("tlab_dns_set_surface_bcs: bad arguments");
        return TLAB_EINVAL;
    }
    try {
        for (int i = 0; i < d->nscal; ++i)
            if ((sfc_jmin[i] != 0 && sfc_jmin[i] != 1) || (sfc_jmax[i] != 0 && sfc_jmax[i] != 1)) throw Fail(TLAB_EINVAL, "SfcType: 0 static or 1 linear");
        d->sfc_jmin.assign(sfc_jmin, sfc_jmin + d->nscal); d->sfc_jmax.assign(sfc_jmax, sfc_jmax + d->nscal);
        d->cpl_jmin.assign(cpl_jmin, cpl_jmin + d->nscal); d->cpl_jmax.assign(cpl_jmax, cpl_jmax + d->nscal);
        d->sref_b.resize(d->nscal, nullptr); d->sref_t.resize(d->nscal, nullptr);
        const size_t pbytes = (size_t)d->nx * d->nz * sizeof(double);
        for (int i = 0; i < d->nscal; ++i) {
            if ((sfc_jmin[i] == 1 || sfc_jmax[i] == 1) && !d->sref_b[i]) {
                hk(hipMalloc((void **)&d->sref_b[i], pbytes), "hipMalloc");
                hk(hipMalloc((void **)&d->sref_t[i], pbytes), "hipMalloc");
            }
        }
        if (!d->sfc_avg) hk(hipMalloc((void **)&d->sfc_avg, sizeof(double)), "hipMalloc");
        return TLAB_OK;
    } catch (const Fail &e) {
        tlab_set_error(e.what());
        return e.code;
    }
}

// nse_eqns == DNS_EQNS_ANELASTIC (tools/dns/rhs_global_incompressible_1.f90:211-214, 275-277, 326-329; physics/opr_burgers.f90:128-183) with the
// background profiles the host's thermodynamics made: rbackground(1:ny), ribackground(1:ny) = 1 / rbackground (HOST pointers; NULL: incompressible)
int tlab_dns_set_anelastic(tlab_dns_t d, const double *rbackground, const double *ribackground) {
    (void)tlab_internal_deferred_flush();
    if (!d) return TLAB_EINVAL;
    try {
        if (!rbackground || !ribackground) ok(tlab_opr_burgers_set_anelastic(0, nullptr, nullptr), "tlab_opr_burgers_set_anelastic");
        else ok(tlab_opr_burgers_set_anelastic(d->ny, rbackground, ribackground), "tlab_opr_burgers_set_anelastic");
        follow_anelastic(d);
        d->anel_owner = d->rb != nullptr;
        return TLAB_OK;
    } catch (const Fail &e) {
        tlab_set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

// [PressureFilter] (operators/opr_filter.f90:46, 78; rhs_global_incompressible_1.f90:286-290): directional 1-D filters applied to p and dp/dy after
// the Poisson solve; NULL = DNS_FILTER_NONE in that direction; repeat may be NULL (1 each).  The filters are not owned.
int tlab_dns_set_pressure_filter(tlab_dns_t d, tlab_filter_t fx, tlab_filter_t fy, tlab_filter_t fz, const int *repeat) {
    (void)tlab_internal_deferred_flush();
    if (!d) return TLAB_EINVAL;
    d->pfilter[0] = fx; d->pfilter[1] = fy; d->pfilter[2] = fz;
    for (int i = 0; i < 3; ++i) d->pfilter_rep[i] = repeat ? repeat[i] : 1;
    return TLAB_OK;
}

int tlab_dns_set_remove_divergence(tlab_dns_t d, int on) {
    (void)tlab_internal_deferred_flush();
    if (!d) return TLAB_EINVAL;
    d->remove_divergence = on != 0;
    return TLAB_OK;
}

int tlab_dns_set_slab(tlab_dns_t d, int koffset) {
    (void)tlab_internal_deferred_flush();
    if (!d || koffset < 0 || koffset + d->nz > d->nz_total) { tlab_set_error("tlab_dns_set_slab: bad offset"); return TLAB_EINVAL; }
    d->koffset = koffset;
    return TLAB_OK;
}

int tlab_time_courant(tlab_dns_t d, double *const *q, double cfla, double cfld, double *pmax, double *dtime) {
    try {
        if (!d || !q || !pmax) throw Fail(TLAB_EINVAL, "tlab_time_courant: bad arguments");
        if (d->dx2i < 0.0) throw Fail(TLAB_EINVAL, "tlab_time_courant: the plans carry no Jacobian (tlab_fdm_plan_set_aux)");
        double mn, mx;
        minmax_impl(d, q[0], q[1], q[2], 1, d->nx, d->ny, d->nz, &mn, &mx);
        pmax[0] = mx;                                   // max of |u|/dx + |v|/dy + |w|/dz over the local box (time.f90:402-451)
        pmax[1] = d->schmidtfactor * d->dx2i;           // time.f90:466
        if (dtime && cfla > 0.0) {                      // time.f90:523-538, explicit RK: min of the two limits
            double dtc = 1.0e300, dtd = 1.0e300;
            if (pmax[0] > 0.0) dtc = cfla / pmax[0];
            if (pmax[1] > 0.0) dtd = cfld / pmax[1];
            *dtime = std::min(dtc, dtd);
        }
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

int tlab_fi_invariant_p(tlab_dns_t d, const double *u, const double *v, const double *w, double *result, double *tmp1) {
    try {
        if (!d || !u || !v || !w || !result || !tmp1) throw Fail(TLAB_EINVAL, "tlab_fi_invariant_p: bad arguments");
        const int nx = d->nx, ny = d->ny, nz = d->nz;
        const long long n = (long long)nx * ny * nz;
        hipStream_t st = tlab_current_stream();
        // result = -((du/dx + dv/dy) + dw/dz), fi_vectorcalculus.f90:130-136
        ok(tlab_opr_partial(1, d->g[0], TLAB_OPR_P1, nx, ny, nz, 0, u, result, nullptr), "OPR_Partial_X");
        ok(tlab_opr_partial(2, d->g[1], TLAB_OPR_P1, nx, ny, nz, 0, v, tmp1, nullptr), "OPR_Partial_Y");
        hk(launch_add1(result, tmp1, n, st), "add");
        ok(tlab_opr_partial(3, d->g[2], TLAB_OPR_P1, nx, ny, nz, 0, w, tmp1, nullptr), "OPR_Partial_Z");
        hk(launch_add1(result, tmp1, n, st), "add");
        hk(launch_negate(result, n, st), "negate");
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

int tlab_minmax(tlab_dns_t d, const double *a, int nx, int ny, int nz, double *amn, double *amx) {
    try {
        if (!d || !a || !amn || !amx) throw Fail(TLAB_EINVAL, "tlab_minmax: bad arguments");
        minmax_impl(d, a, nullptr, nullptr, 0, nx, ny, nz, amn, amx);
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    }
}

int tlab_dns_begin_step(tlab_dns_t d) {
    (void)tlab_internal_deferred_flush();
    if (!d) return TLAB_EINVAL;
    d->fresh = true;
    return TLAB_OK;
}

// ---- which allocations play q, s, hq, hs, txc (include/tlab_amd.h: tlab_dns_place_arrays) ----
// A kernel that streams many arrays at once runs at a rate that depends on WHICH device allocations they are -- not on any one of them (each alone
// reads / writes at the same rate), on the set: tools/placement_probe measures 4.8 .. 5.9 TB/s for one 13-stream kernel over sets drawn from a pool
// of 1-GiB hipMalloc allocations, reproducibly per set (profiles/r05/placement_*.txt).  It is what made "the slow state of the box" of rounds 3-5:
// the same binary, the same box, 16.0 .. 17.2 ms per substep from process to process.  Neither the virtual alignment nor a skew between the arrays
// of one allocation predicts it, so the assignment is searched: time the substep itself on candidate assignments and keep the fastest.
namespace {
// the tools of the two placement searches below: one Runge-Kutta step of three substeps (the first on fresh tendencies, the others accumulating and
// scaling: the kernels of a real step) timed by events on the library's stream, after one untimed substep (first touch of the arrays)
struct PlaceTimer {
    tlab_dns_t d;
    double dtime;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool fresh_on_entry;
    PlaceTimer(tlab_dns_t d_, double dtime_) : d(d_), dtime(dtime_), st(tlab_current_stream()), fresh_on_entry(d_->fresh) {
        hk(hipEventCreate(&e0), "hipEventCreate");
        hk(hipEventCreate(&e1), "hipEventCreate");
    }
    ~PlaceTimer() {      // released on every way out (a failing trial throws); the caller's begin_step flag survives the trial substeps
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        d->fresh = fresh_on_entry;
    }
    double step(double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc, bool first_touch) {
        const double kdt[3] = {1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0}, kco[3] = {-5.0 / 9.0, -153.0 / 128.0, 1.0};
        if (first_touch) {
            d->fresh = true;
            rhs_impl(d, dtime * kdt[0], q, s, hq, hs, txc, true, kco[0], 1);
        }
        hk(hipEventRecord(e0, st), "hipEventRecord");
        d->fresh = true;
        for (int k = 0; k < 3; ++k) rhs_impl(d, dtime * kdt[k], q, s, hq, hs, txc, true, kco[k], k < 2);
        hk(hipEventRecord(e1, st), "hipEventRecord");
        hk(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0.0f;
        hk(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        return (double)ms / 3.0;
    }
};
struct XorShift {      // the same sequence everywhere
    unsigned long long x;
    explicit XorShift(unsigned seed) : x(0x9E3779B97F4A7C15ull ^ ((unsigned long long)seed * 0xBF58476D1CE4E5B9ull + 1ull)) {}
    int operator()(int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (unsigned long long)m); }
};
// The search keeps the minimum of single timings, which is biased by noise (and the first assignment is timed straight after the warm-up): the
// winner and the first assignment are timed again, back to back, three times each; the winner stays only if its median is below the other's.
// Returns (ms of the first assignment, ms of the assignment kept) from those repeats.
std::pair<double, double> confirm(const std::function<double(const std::vector<int> &)> &trial, const std::vector<int> &ident, std::vector<int> &best) {
    auto med3 = [](double a, double b, double c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); };
    if (best == ident) {
        const double m = med3(trial(ident), trial(ident), trial(ident));
        return {m, m};
    }
    double a[3], b[3];
    for (int r = 0; r < 3; ++r) { a[r] = trial(ident); b[r] = trial(best); }
    const double mi = med3(a[0], a[1], a[2]), mb = med3(b[0], b[1], b[2]);
    if (!(mb < mi)) { best = ident; return {mi, mi}; }
    return {mi, mb};
}
}      // namespace

int tlab_dns_place_arrays(tlab_dns_t d, int npool, double *const *pool, const double *const *state, double dtime, int random_trials, unsigned seed,
                          int *assignment, double *report) {
    try {
        if (!d || !pool || !assignment || dtime <= 0.0 || random_trials < 0) throw Fail(TLAB_EINVAL, "tlab_dns_place_arrays: bad arguments");
        const int ns = d->nscal, nroles = 2 * (3 + ns) + 9;
        if (npool < nroles) throw Fail(TLAB_EINVAL, "tlab_dns_place_arrays: the pool must hold at least 2 (3 + nscal) + 9 arrays");
        for (int i = 0; i < npool; ++i) {
            if (!pool[i]) throw Fail(TLAB_EINVAL, "tlab_dns_place_arrays: null array in the pool");
            for (int j = 0; j < i; ++j)
                if (pool[j] == pool[i]) throw Fail(TLAB_EINVAL, "tlab_dns_place_arrays: the same array twice in the pool");
        }
        PlaceTimer T(d, dtime);
        hipStream_t st = T.st;
        const size_t fbytes = (size_t)d->nx * d->ny * d->nz * sizeof(double);
        auto trial = [&](const std::vector<int> &a) {
            std::vector<double *> q(3), s((size_t)std::max(ns, 1)), hq(3), hs((size_t)std::max(ns, 1)), txc(9);
            int r = 0;
            for (int i = 0; i < 3; ++i) q[i] = pool[a[r++]];
            for (int i = 0; i < ns; ++i) s[i] = pool[a[r++]];
            for (int i = 0; i < 3; ++i) hq[i] = pool[a[r++]];
            for (int i = 0; i < ns; ++i) hs[i] = pool[a[r++]];
            for (int i = 0; i < 9; ++i) txc[i] = pool[a[r++]];
            for (int i = 0; i < 3 + ns; ++i) {
                double *dst = i < 3 ? q[i] : s[i - 3];
                if (state && state[i]) hk(hipMemcpyAsync(dst, state[i], fbytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
                else hk(hipMemsetAsync(dst, 0, fbytes, st), "hipMemsetAsync");
            }
            return T.step(q.data(), s.data(), hq.data(), hs.data(), txc.data(), true);
        };
        XorShift rnd(seed);
        std::vector<int> ident((size_t)nroles);
        for (int i = 0; i < nroles; ++i) ident[i] = i;
        std::vector<double> all;
        std::vector<int> best = ident;
        const double first = trial(ident);
        double bestms = first;
        all.push_back(first);
        for (int t = 0; t < random_trials; ++t) {
            std::vector<int> perm((size_t)npool);
            for (int i = 0; i < npool; ++i) perm[i] = i;
            for (int i = npool - 1; i > 0; --i) std::swap(perm[i], perm[rnd(i + 1)]);
            perm.resize((size_t)nroles);
            const double ms = trial(perm);
            all.push_back(ms);
            if (ms < bestms) { bestms = ms; best = perm; }
        }
        // one pass of single-role exchanges against the unused arrays (skipped when the pool has none to spare)
        if (npool > nroles && random_trials > 0) {
            for (int r = 0; r < nroles; ++r) {
                std::vector<char> used((size_t)npool, 0);
                for (int v : best) used[v] = 1;
                int cand = -1;
                for (int tries = 0; tries < 4 * npool && cand < 0; ++tries) { const int c = rnd(npool); if (!used[c]) cand = c; }
                if (cand < 0) continue;
                std::vector<int> a = best;
                a[r] = cand;
                const double ms = trial(a);
                all.push_back(ms);
                if (ms < bestms * 0.999) { bestms = ms; best = a; }
            }
        }
        double ms_first = first, ms_kept = bestms;
        if (random_trials > 0) { auto c = confirm(trial, ident, best); ms_first = c.first; ms_kept = c.second; }
        for (int r = 0; r < nroles; ++r) assignment[r] = best[r];
        if (report) {
            std::vector<double> sorted = all;
            std::sort(sorted.begin(), sorted.end());
            report[0] = ms_first; report[1] = ms_kept; report[2] = sorted[sorted.size() / 2]; report[3] = sorted.back(); report[4] = (double)all.size();
        }
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

// ---- the same question for a Tlab HOST (include/tlab_amd.h: tlab_dns_place_blocks) ----
// The host's arrays are two-dimensional: q(isize_field, 3), s(isize_field, ns), hq, hs, txc(isize_txc_field, 9) (base/tlab_memory.f90:201-207,
// dns_main.f90:103-104) -- the components of a block lie a fixed stride apart and cannot be placed one by one.  What CAN be chosen is which
// allocation plays each block: the host allocates ncand candidates per block (index 0 = the ones it holds), the substep is timed on combinations,
// and the host re-associates its pointers with the winners before any field is read.
int tlab_dns_place_blocks(tlab_dns_t d, int ncand, double *const *cand_q, double *const *cand_s, double *const *cand_hq, double *const *cand_hs,
                          double *const *cand_txc, long long txc_stride, double dtime, int random_trials, unsigned seed, int *choice, double *report) {
    try {
        if (!d || ncand < 1 || !cand_q || !cand_hq || !cand_txc || !choice || dtime <= 0.0 || random_trials < 0)
            throw Fail(TLAB_EINVAL, "tlab_dns_place_blocks: bad arguments");
        const int ns = d->nscal;
        const long long n = (long long)d->nx * d->ny * d->nz;
        if (ns > 0 && (!cand_s || !cand_hs)) throw Fail(TLAB_EINVAL, "tlab_dns_place_blocks: candidates for s, hs are missing");
        if (txc_stride < n) throw Fail(TLAB_EINVAL, "tlab_dns_place_blocks: txc_stride below the field size");
        double *const *cands[5] = {cand_q, ns > 0 ? cand_s : nullptr, cand_hq, ns > 0 ? cand_hs : nullptr, cand_txc};      // (no scalars: whatever stands there is not looked at)
        for (int b = 0; b < 5; ++b) {
            if (!cands[b]) continue;
            for (int i = 0; i < ncand; ++i) {
                if (!cands[b][i]) throw Fail(TLAB_EINVAL, "tlab_dns_place_blocks: null candidate");
                for (int b2 = 0; b2 <= b; ++b2)
                    for (int j = 0; cands[b2] && j < (b2 == b ? i : ncand); ++j)
                        if (cands[b2][j] == cands[b][i]) throw Fail(TLAB_EINVAL, "tlab_dns_place_blocks: the same array twice among the candidates");
            }
        }
        PlaceTimer T(d, dtime);
        hipStream_t st = T.st;
        std::vector<char> touched((size_t)5 * ncand, 0);
        auto trial = [&](const std::vector<int> &a) {
            std::vector<double *> q(3), s((size_t)std::max(ns, 1)), hq(3), hs((size_t)std::max(ns, 1)), txc(9);
            for (int i = 0; i < 3; ++i) { q[i] = cand_q[a[0]] + (long long)i * n; hq[i] = cand_hq[a[2]] + (long long)i * n; }
            for (int i = 0; i < ns; ++i) { s[i] = cand_s[a[1]] + (long long)i * n; hs[i] = cand_hs[a[3]] + (long long)i * n; }
            for (int i = 0; i < 9; ++i) txc[i] = cand_txc[a[4]] + (long long)i * txc_stride;
            // the trial fields are zeros (the host has read nothing yet; the kernels' time does not depend on the values)
            hk(hipMemsetAsync(q[0], 0, (size_t)3 * n * sizeof(double), st), "hipMemsetAsync");
            if (ns > 0) hk(hipMemsetAsync(s[0], 0, (size_t)ns * n * sizeof(double), st), "hipMemsetAsync");
            bool first_touch = false;
            for (int b = 0; b < 5; ++b) { char &t = touched[(size_t)b * ncand + a[b]]; first_touch = first_touch || !t; t = 1; }
            return T.step(q.data(), s.data(), hq.data(), hs.data(), txc.data(), first_touch);
        };
        XorShift rnd(seed);
        const std::vector<int> ident(5, 0);
        std::vector<int> best = ident;
        std::vector<double> all;
        const double first = trial(ident);
        double bestms = first;
        all.push_back(first);
        if (ncand > 1) {
            for (int t = 0; t < random_trials; ++t) {
                std::vector<int> a(5);
                for (int b = 0; b < 5; ++b) a[b] = rnd(ncand);
                const double ms = trial(a);
                all.push_back(ms);
                if (ms < bestms) { bestms = ms; best = a; }
            }
            if (random_trials > 0)      // one pass over the blocks: every other candidate of one block with the rest held
                for (int b = 0; b < 5; ++b) {
                    if (!cands[b]) continue;
                    for (int c = 0; c < ncand; ++c) {
                        if (c == best[b]) continue;
                        std::vector<int> a = best;
                        a[b] = c;
                        const double ms = trial(a);
                        all.push_back(ms);
                        if (ms < bestms * 0.999) { bestms = ms; best = a; }
                    }
                }
        }
        double ms_first = first, ms_kept = bestms;
        if (ncand > 1 && random_trials > 0) { auto c = confirm(trial, ident, best); ms_first = c.first; ms_kept = c.second; }
        for (int b = 0; b < 5; ++b) choice[b] = best[b];
        if (report) {
            std::vector<double> sorted = all;
            std::sort(sorted.begin(), sorted.end());
            report[0] = ms_first; report[1] = ms_kept; report[2] = sorted[sorted.size() / 2]; report[3] = sorted.back(); report[4] = (double)all.size();
        }
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.what());
        return f.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

int tlab_dns_set_fusion(tlab_dns_t d, int on) {
    (void)tlab_internal_deferred_flush();
    if (!d) return TLAB_EINVAL;
    d->fuse = on != 0;
    return TLAB_OK;
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
int tlab_pw_fill(double *a, double value, long long n) { PW_GUARD(launch_fill(a, value, n, tlab_current_stream())) }
int tlab_pw_scale(double *a, double alpha, long long n) { PW_GUARD(launch_scale(a, alpha, n, tlab_current_stream())) }
int tlab_pw_final_update(double *q, double *h, const double *g, const double *pb, const double *pt, double dte, double kco, int scale, int nx, int ny,
                         int nz) { PW_GUARD(launch_final_update(q, h, g, pb, pt, dte, kco, scale, nx, ny, nz, tlab_current_stream())) }
int tlab_pencil_repack(double *slab, double *buffer, int nxh, int ny, int kmax, int nproc, const int *ioff, int dir) {
    if (!slab || !buffer || !ioff || nproc < 1 || nproc > 8 || nxh < nproc) { tlab_set_error("tlab_pencil_repack: bad arguments (1..8 peers)"); return TLAB_EINVAL; }
    PW_GUARD(launch_pencil_repack(slab, buffer, nxh, ny, kmax, nproc, ioff, nullptr, dir, tlab_current_stream()))
}
int tlab_pencil_repack_blocks(double *slab, double *buffer, int nxh, int ny, int kmax, int nblocks, const int *start, const long long *base, int dir) {
    if (!slab || !buffer || !start || !base || nblocks < 1 || nblocks > 16 || start[0] != 0) { tlab_set_error("tlab_pencil_repack_blocks: bad arguments (1..16 blocks, the first at kx = 0)"); return TLAB_EINVAL; }
    for (int p = 0; p + 1 < nblocks; ++p)
        if (start[p + 1] < start[p] || start[p + 1] > nxh) { tlab_set_error("tlab_pencil_repack_blocks: block starts must increase within [0, nx/2+1]"); return TLAB_EINVAL; }
    PW_GUARD(launch_pencil_repack(slab, buffer, nxh, ny, kmax, nblocks, start, base, dir, tlab_current_stream()))
}
int tlab_pw_get_wall_planes(const double *f, double *hb, double *ht, int nx, int ny, int nz) { PW_GUARD(launch_get_wall_planes(f, hb, ht, nx, ny, nz, tlab_current_stream())) }
int tlab_pw_fill_wall_planes(double *f, double vb, double vt, int nx, int ny, int nz) { PW_GUARD(launch_fill_wall_planes(f, vb, vt, nx, ny, nz, tlab_current_stream())) }
int tlab_pw_set_wall_planes(double *f, const double *pb, const double *pt, int nx, int ny, int nz) { PW_GUARD(launch_set_wall_planes(f, pb, pt, nx, ny, nz, tlab_current_stream())) }

}  // extern "C"
