// RHS_GLOBAL_INCOMPRESSIBLE_1 / TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT on z-slabs (ims_npro_i x ims_npro_k = 1 x P), native driver.
//
// Reference: the same operator list as rhs.cpp (tools/dns/rhs_global_incompressible_1.f90:98-398) with the MPI branches of the z-operators --
// OPR_Partial_Z (operators/opr_partial.f90:185-195, 248-253), OPR_Burgers_Z (physics/opr_burgers.f90:386-426), OPR_Fourier_Z_Forward / _Backward
// (operators/opr_fourier.f90:343-428), each of which wraps its 1-D work in TLabMPI_Trp_ExecK_Forward / _Backward (base/tlab_mpi_transpose.f90:343-458).
// Here no field is transposed for a derivative:
//   * z-derivatives: the compact systems are partitioned at the slab boundaries (zslab.hip).  An operator needs 3 halo planes of its operand and,
//     per line and implicit system, one value from each ring neighbour ("head" to the left, "tail" to the right).  The z-terms come last in every
//     equation, so that those messages travel on the transport's stream while the x / y operators run on the compute stream.
//   * OPR_Poisson: z-slab -> kx-pencil with ONE all-to-all after the x-FFT (z-FFT, per-mode solves and inverse z-FFT are local on the pencil) and one
//     per output field on the way back; every rank's kx range is cut in two halves with a plan each, and the six half-size all-to-alls queue up as
//     fwd A, fwd B, back p A, back dp A, back p B, back dp B: the solves of A run under fwd B, those of B under the returns of A.
// The exchanges go through a tlab_slab_transport (include/tlab_amd.h): RCCL (comm.hip), the single-process loopback below, or the caller's own.
// This file holds host logic only; every kernel is reached through the C ABI of the operator library, as a Fortran host would reach it.
#include "../../include/tlab_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

extern hipStream_t tlab_current_stream();
int tlab_internal_deferred_flush();      // deferred.cpp
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();
// zslab.hip: the z-slab operators with the neighbours' halo planes of every operand in buffers of their own ({lo, hi}, 3 planes each)
bool tlab_internal_poisson_has_own_x(tlab_poisson_plan_t P);      // poisson.hip
int tlab_internal_zslab_partial_z(tlab_zslab_plan_t P, int phase, int nx, int ny, const double *u, const double *const *u_halo, const double *ub,
                                  const double *const *ub_halo, double scale, double *head, double *tail, const double *tail_left,
                                  const double *head_right, double *result, int acc);
int tlab_internal_zslab_burgers_z_n(tlab_zslab_plan_t P, int phase, int nx, int ny, int nf, const double *nu, const double *const *s,
                                    const double *const *s_lo, const double *const *s_hi, const double *vel, double *head, double *tail,
                                    const double *tail_left, const double *head_right, double *const *result, int acc, const int *fin, double dte,
                                    double kco, int scale);
int tlab_internal_zslab_gradient_final_z(tlab_zslab_plan_t P, int nx, int ny, const double *p, const double *const *p_halo, const double *tail_left,
                                         const double *head_right, double *q, double *h, double dte, double kco, int scale);

extern "C" bool tlab_internal_anelastic();       // capi.cpp: the operator state set by tlab_opr_burgers_set_anelastic / _set_dealiasing
extern "C" bool tlab_internal_dealiasing();

extern "C" int tlab_internal_dns_neumann_weights(tlab_dns_t d, int ibc, const double **w, int *K);      // rhs.cpp

namespace tlab {
hipError_t launch_wall_weighted(const double *a1, const double *a2, const double *wb, const double *wt, int K, double *ob1, double *ot1, double *ob2,
                                double *ot2, int nx, int ny, int nz, hipStream_t st);                                      // pointwise.hip
hipError_t launch_wall_fix(double *q, double *h, const double *sb, const double *st, double dte, double kco, int scale, int nx, int ny, int nz,
                           hipStream_t stream);
hipError_t launch_copy_blocks(int n, const double *const *src, double *const *dst, const long long *cnt, hipStream_t st);      // pointwise.hip
hipError_t launch_plane_avg(const double *t, int j, int nx, int ny, int nz, double *avg, hipStream_t st);
hipError_t launch_surface_flux_avg(double *ref, const double *t, int j, double sign, double diff, double cpl, double avg, int nx, int ny, int nz,
                                   hipStream_t st);
hipError_t launch_get_wall_planes(const double *f, double *hb, double *ht, int nx, int ny, int nz, hipStream_t st);
}

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
double *dalloc(size_t ndoubles) {
    double *p = nullptr;
    hk(hipMalloc((void **)&p, std::max<size_t>(ndoubles, 1) * sizeof(double)), "hipMalloc");
    hk(hipMemset(p, 0, std::max<size_t>(ndoubles, 1) * sizeof(double)), "hipMemset");
    return p;
}

constexpr int HALO = 3;      // planes each side: the 7-diagonal right-hand side of the second derivative reaches 3 rows

// ---- single-process loopback transport: every exchange is a set of device copies on the caller's stream ----
struct Loopback {
    int P;
};
// (the copies of one exchange go out as a few batched launches: dozens of hipMemcpyAsync calls per exchange made the diagnostic host-bound)
int lb_ring(void *ctx, void *stream, int nmsg, const long long *count, double *const *to_left, double *const *to_right, double *const *from_right,
            double *const *from_left) {
    const int P = static_cast<Loopback *>(ctx)->P;
    hipStream_t st = (hipStream_t)stream;
    std::vector<const double *> src;
    std::vector<double *> dst;
    std::vector<long long> cnt;
    for (int r = 0; r < P; ++r) {
        const int left = (r + P - 1) % P, right = (r + 1) % P;
        for (int i = 0; i < nmsg; ++i) {
            src.push_back(to_left[r * nmsg + i]); dst.push_back(from_right[left * nmsg + i]); cnt.push_back(count[i]);
            src.push_back(to_right[r * nmsg + i]); dst.push_back(from_left[right * nmsg + i]); cnt.push_back(count[i]);
        }
    }
    return tlab::launch_copy_blocks((int)src.size(), src.data(), dst.data(), cnt.data(), st) == hipSuccess ? 0 : TLAB_EHIP;
}
int lb_a2a(void *ctx, void *stream, double *const *send, const long long *scount, double *const *recv, const long long *rcount) {
    const int P = static_cast<Loopback *>(ctx)->P;
    hipStream_t st = (hipStream_t)stream;
    std::vector<const double *> sp;
    std::vector<double *> dp;
    std::vector<long long> cn;
    for (int dst = 0; dst < P; ++dst) {
        long long ro = 0;
        for (int src = 0; src < P; ++src) {
            long long so = 0;
            for (int p = 0; p < dst; ++p) so += scount[src * P + p];
            const long long cnt = scount[src * P + dst];
            if (cnt != rcount[dst * P + src]) return TLAB_EINVAL;
            if (cnt > 0) { sp.push_back(send[src] + so); dp.push_back(recv[dst] + ro); cn.push_back(cnt); }
            ro += cnt;
        }
    }
    return tlab::launch_copy_blocks((int)sp.size(), sp.data(), dp.data(), cn.data(), st) == hipSuccess ? 0 : TLAB_EHIP;
}
int lb_wait(void *, void *, int) { return TLAB_OK; }
int lb_allreduce(void *ctx, double *v, int n, int op) {
    const int P = static_cast<Loopback *>(ctx)->P;
    for (int i = 0; i < n; ++i) {
        double a = v[i];
        for (int r = 1; r < P; ++r) a = op == 0 ? std::max(a, v[r * n + i]) : (op == 1 ? std::min(a, v[r * n + i]) : a + v[r * n + i]);
        for (int r = 0; r < P; ++r) v[r * n + i] = a;
    }
    return TLAB_OK;
}
void lb_destroy(void *ctx) { delete static_cast<Loopback *>(ctx); }

// one local rank
struct Rank {
    int r = 0;                                   // ims_pro_k
    tlab_zslab_plan_t zplan = nullptr;
    tlab_poisson_plan_t poisson = nullptr, poisson_b = nullptr;
    tlab_dns_t dns = nullptr;                    // monitors (TIME_COURANT, MINMAX), made on first use
    std::vector<double *> sref_b, sref_t;        // BcsScalJmin / Jmax%ref(:,:,is) of the scalars with a dynamic surface (boundary_bcs.f90:76-87)
    double *sfc_avg = nullptr;                   // one double: plane average
    double *hb = nullptr, *ht = nullptr;         // BcsFlowJmin/Jmax%ref(:,:,2): Neumann data of the pressure, wall planes of the Neumann fields
    double *head = nullptr, *tail = nullptr, *head_right = nullptr, *tail_left = nullptr;   // interface values, [2 (3 + ns)][nx*ny]
    double *pen[3] = {nullptr, nullptr, nullptr};   // complex kx-pencils (nxl, ny, nz_total)
    double *pack[2] = {nullptr, nullptr};           // complex slabs blocked by peer
    // The neighbours' halo planes (3 before = lo, 3 after = hi) of the fields a z-operator reads: q(1:3), s(1:ns), hq(3), tmp1.  Buffers of the
    // driver's own: the module arrays of a Fortran host (q(isize_field, 3): columns back to back) have no room around a field.
    double *halo = nullptr;                         // one allocation: [3 + ns + 2][2][3 planes]
    std::vector<const double *> lo, hi;             // per field slot: 0..2 q, 3..2+ns s, 3+ns hq(3), 4+ns tmp1
    std::vector<double *> q, s, hq, hs, txc;     // bound module arrays
    bool bound = false;
};

}  // namespace

struct tlab_slab_dns {
    tlab_slab_transport tr{};
    tlab_fdm_plan_t g[3] = {nullptr, nullptr, nullptr}, gy_elliptic = nullptr;
    int P = 1, nx = 0, ny = 0, nzt = 0, kmax = 0, nxh = 0, nscal = 0, stages = 1;
    long long npage = 0, n = 0;
    double visc = 0.0;
    std::vector<double> schmidt;
    std::vector<int> nxl, ioff, nxa;
    // the slab <-> pencil repack folded into the library's own x-transforms (tlab_poisson_fft_x_packed), and v finished by the inverse transform of
    // dp^/dy; TLAB_SLAB_FUSED_X=0 keeps the separate passes (then the results equal the Python driver's to the bit: rocFFT's inverse)
    bool fused_x = false;
    struct VFinal { bool armed = false; double dte = 0.0, kco = 0.0; int scale = 0; } vf;
    std::vector<int> sg_start;                   // one-piece block map
    std::vector<long long> sg_base;
    std::vector<int> st_start;                   // two-stage block map (tlab_pencil_repack_blocks)
    std::vector<long long> st_base;
    long long st_split = 0;                      // doubles of the A part of a pack buffer
    int flow_jmin[3] = {TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET};
    int flow_jmax[3] = {TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET};
    std::vector<int> scal_jmin, scal_jmax;
    bool fresh = false;
    std::vector<int> sfc_jmin, sfc_jmax;          // BcsScalJmin / Jmax%SfcType (0 static, 1 linear) and %cpl per scalar
    std::vector<double> cpl_jmin, cpl_jmax;
    bool remove_divergence = true;     // [Main] TermDivergence: forcing div(hq + q/dte) (rhs_global_incompressible_1.f90:177-232); false: div(hq) (:234-250)
    std::vector<Rank> rk;
    ~tlab_slab_dns() {
        for (Rank &R : rk) {
            if (R.dns) (void)tlab_dns_destroy(R.dns);
            if (R.poisson) (void)tlab_poisson_plan_destroy(R.poisson);
            if (R.poisson_b) (void)tlab_poisson_plan_destroy(R.poisson_b);
            if (R.zplan) (void)tlab_zslab_plan_destroy(R.zplan);
            for (double *p : {R.hb, R.ht, R.head, R.tail, R.head_right, R.tail_left, R.pen[0], R.pen[1], R.pen[2], R.pack[0], R.pack[1], R.halo, R.sfc_avg})
                if (p) (void)hipFree(p);
            for (double *p : R.sref_b) if (p) (void)hipFree(p);
            for (double *p : R.sref_t) if (p) (void)hipFree(p);
        }
        if (tr.destroy) tr.destroy(tr.ctx);
    }
};

namespace {

using D = tlab_slab_dns;

void need_bound(D *d) {
    for (const Rank &R : d->rk)
        if (!R.bound) throw Fail(TLAB_EINVAL, "tlab_slab_dns: the arrays of every local rank must be bound first (tlab_slab_dns_bind)");
}
void tck(int rc, const char *what) {
    if (rc < 0) throw Fail(rc, std::string("slab transport: ") + what + " failed");
}
void twait(D *d, int ticket) { tck(d->tr.wait(d->tr.ctx, (void *)tlab_current_stream(), ticket), "wait"); }

// One ring exchange of nmsg messages per local rank; the four pointer tables are filled by `fill(R, l, to_left, to_right, from_right, from_left)`.
template <class F>
int ring(D *d, int nmsg, const std::vector<long long> &count, F fill) {
    const int L = (int)d->rk.size();
    std::vector<double *> tl((size_t)L * nmsg), trr((size_t)L * nmsg), fr((size_t)L * nmsg), fl((size_t)L * nmsg);
    for (int l = 0; l < L; ++l) fill(d->rk[l], &tl[(size_t)l * nmsg], &trr[(size_t)l * nmsg], &fr[(size_t)l * nmsg], &fl[(size_t)l * nmsg]);
    const int t = d->tr.ring_start(d->tr.ctx, (void *)tlab_current_stream(), nmsg, count.data(), tl.data(), trr.data(), fr.data(), fl.data());
    tck(t, "ring_start");
    return t;
}

// Halo planes of `nf` fields (picked by `field(R, i)` -> (array, halo slot)): my last planes go to the right neighbour's planes -H..-1 (its lo
// buffer), my first planes to the left neighbour's planes kmax..kmax+H-1 (its hi buffer); periodic in z.  The send side is the field itself.
struct Slot {
    double *f;
    int slot;
};
template <class F>
int halo_start(D *d, int nf, F field) {
    const long long Hn = HALO * d->npage, n = d->n;
    std::vector<long long> count((size_t)nf, Hn);
    return ring(d, nf, count, [&](Rank &R, double **tl, double **tr, double **fr, double **fl) {
        for (int i = 0; i < nf; ++i) {
            const Slot sl = field(R, i);
            tl[i] = sl.f;
            tr[i] = sl.f + n - Hn;
            fr[i] = const_cast<double *>(R.hi[sl.slot]);       // planes kmax .. kmax+H-1
            fl[i] = const_cast<double *>(R.lo[sl.slot]);       // planes -H .. -1
        }
    });
}

// head -> left neighbour (its head_right), tail -> right neighbour (its tail_left); nrows line-sets of nx*ny
int msg_start(D *d, int nrows) {
    std::vector<long long> count(1, (long long)nrows * d->npage);
    return ring(d, 1, count, [&](Rank &R, double **tl, double **tr, double **fr, double **fl) {
        tl[0] = R.head; tr[0] = R.tail; fr[0] = R.head_right; fl[0] = R.tail_left;
    });
}

struct Eq {
    double *f, *h;
    double nu;
};
std::vector<Eq> eqs(D *d, Rank &R) {
    std::vector<Eq> E;
    for (int i = 0; i < 3; ++i) E.push_back({R.q[i], R.hq[i], d->visc});
    for (int i = 0; i < d->nscal; ++i) E.push_back({R.s[i], R.hs[i], d->visc / d->schmidt[i]});
    return E;
}

// hq, hs += Burgers_dir of every transported field, four per launch (they share the advecting velocity q_dir)
void badd_all(D *d, Rank &R, int dir, bool overwrite) {
    const std::vector<Eq> E = eqs(d, R);
    for (size_t e0 = 0; e0 < E.size(); e0 += 4) {
        const int nf = (int)std::min<size_t>(4, E.size() - e0);
        double nus[4];
        const double *sp[4];
        double *rp[4];
        for (int f = 0; f < nf; ++f) { nus[f] = E[e0 + f].nu; sp[f] = E[e0 + f].f; rp[f] = E[e0 + f].h; }
        ok(tlab_opr_burgers_add_n(dir, d->g[dir - 1], d->nx, d->ny, d->kmax, 0, nf, nus, sp, R.q[dir - 1], rp, R.txc[6], R.txc[7], overwrite ? 1 : 0),
           "tlab_opr_burgers_add_n");
    }
}
// fin (phase 2): equations whose tendency is complete with the z term and whose walls are Dirichlet (the scalars on the last pass of a substep)
// take their Runge-Kutta update in the epilogue of the kernel
void zburgers_all(D *d, Rank &R, int phase, const std::vector<int> *fin = nullptr, double dte = 0.0, double kco = 1.0, int scale = 0) {
    const std::vector<Eq> E = eqs(d, R);
    for (size_t e0 = 0; e0 < E.size(); e0 += 4) {
        const int nf = (int)std::min<size_t>(4, E.size() - e0);
        double nus[4];
        const double *sp[4];
        double *rp[4];
        int fl[4] = {0, 0, 0, 0};
        for (int f = 0; f < nf; ++f) {
            nus[f] = E[e0 + f].nu; sp[f] = E[e0 + f].f; rp[f] = E[e0 + f].h;
            if (fin && phase == 2) fl[f] = (*fin)[e0 + f];
        }
        const long long o = 2 * (long long)e0 * d->npage;
        ok(tlab_internal_zslab_burgers_z_n(R.zplan, phase, d->nx, d->ny, nf, nus, sp, &R.lo[e0], &R.hi[e0], phase == 2 ? R.q[2] : nullptr, R.head + o,
                                           R.tail + o, R.tail_left + o, R.head_right + o, phase == 2 ? rp : nullptr, 1, fl, dte, kco, scale),
           "tlab_zslab_burgers_z_n");
    }
}
void padd(D *d, Rank &R, int dir, const double *u, const double *ub, double scale, double *res, int acc) {
    ok(tlab_opr_partial_add(dir, d->g[dir - 1], d->nx, d->ny, d->kmax, 0, u, ub, scale, res, acc, R.txc[6], R.txc[7]), "tlab_opr_partial_add");
}
// us, ubs: halo slots of the operands (ubs ignored without ub)
void zpartial(D *d, Rank &R, int phase, const double *u, int us, const double *ub, int ubs, double scale, double *res, int acc) {
    const double *uh[2] = {R.lo[us], R.hi[us]}, *ubh[2] = {R.lo[ub ? ubs : us], R.hi[ub ? ubs : us]};
    ok(tlab_internal_zslab_partial_z(R.zplan, phase, d->nx, d->ny, u, uh, ub, ubh, scale, R.head, R.tail, R.tail_left, R.head_right, res, acc),
       "tlab_zslab_partial_z");
}

// ---- OPR_Poisson on kx-pencils: forcing in tmp1, Neumann data in hb / ht; p -> tmp1, dp/dy -> tmp3 ----
int a2a(D *d, const std::vector<double *> &send, const std::vector<long long> &scnt, const std::vector<double *> &recv, const std::vector<long long> &rcnt) {
    const int t = d->tr.alltoallv_start(d->tr.ctx, (void *)tlab_current_stream(), send.data(), scnt.data(), recv.data(), rcnt.data());
    tck(t, "alltoallv_start");
    return t;
}

void repack(D *d, double *slab, double *buf, int dir) {
    const int P = d->P;
    if (P <= 8) {
        ok(tlab_pencil_repack(slab, buf, d->nxh, d->ny, d->kmax, P, d->ioff.data(), dir), "tlab_pencil_repack");
    } else {    // up to 16 peers through the block form
        std::vector<long long> base((size_t)P);
        long long off = 0;
        for (int p = 0; p < P; ++p) { base[p] = off; off += (long long)d->nxl[p] * d->ny * d->kmax; }
        ok(tlab_pencil_repack_blocks(slab, buf, d->nxh, d->ny, d->kmax, P, d->ioff.data(), base.data(), dir), "tlab_pencil_repack_blocks");
    }
}

// the x-transforms of the fused route: real slab <-> pack buffer; the one of dp^/dy finishes v when the RHS armed it
void x_to_pack(D *d, Rank &R, double *real, double *pack) {
    const bool st = d->stages == 2;
    ok(tlab_poisson_fft_x_packed(R.poisson, 1, real, pack, st ? 2 * d->P : d->P, st ? d->st_start.data() : d->sg_start.data(),
                                 st ? d->st_base.data() : d->sg_base.data()), "tlab_poisson_fft_x_packed");
}
void x_from_pack(D *d, Rank &R, double *pack, double *real, bool dpdy) {
    const bool st = d->stages == 2;
    const int nb = st ? 2 * d->P : d->P;
    const int *start = st ? d->st_start.data() : d->sg_start.data();
    const long long *base = st ? d->st_base.data() : d->sg_base.data();
    if (dpdy && d->vf.armed)
        ok(tlab_poisson_fft_x_packed_final(R.poisson, pack, R.q[1], R.hq[1], d->vf.dte, d->vf.kco, d->vf.scale, nb, start, base), "tlab_poisson_fft_x_packed_final");
    else ok(tlab_poisson_fft_x_packed(R.poisson, -1, pack, real, nb, start, base), "tlab_poisson_fft_x_packed");
}

// slab <-> pencil exchange of one complex field: slab side = pack buffer blocked by peer, pencil side = (nxl_r, ny, nz_total)
int pencil_exchange(D *d, int i_pen, int i_pack, bool forward) {
    const int P = d->P, L = (int)d->rk.size();
    std::vector<double *> send(L), recv(L);
    std::vector<long long> scnt((size_t)L * P), rcnt((size_t)L * P);
    for (int l = 0; l < L; ++l) {
        Rank &R = d->rk[l];
        for (int p = 0; p < P; ++p) {
            const long long slab_cnt = 2LL * d->nxl[p] * d->ny * d->kmax, pen_cnt = 2LL * d->nxl[R.r] * d->ny * d->kmax;
            scnt[(size_t)l * P + p] = forward ? slab_cnt : pen_cnt;
            rcnt[(size_t)l * P + p] = forward ? pen_cnt : slab_cnt;
        }
        send[l] = forward ? R.pack[i_pack] : R.pen[i_pen];
        recv[l] = forward ? R.pen[i_pen] : R.pack[i_pack];
    }
    return a2a(d, send, scnt, recv, rcnt);
}

void poisson_pencil_single(D *d) {
    const bool direct = d->gy_elliptic != nullptr;
    for (Rank &R : d->rk) {
        ok(tlab_poisson_set_wall_planes(R.poisson, R.txc[0], R.hb, R.ht), "tlab_poisson_set_wall_planes");
        if (d->fused_x) { x_to_pack(d, R, R.txc[0], R.pack[0]); continue; }
        ok(tlab_poisson_fft_x(R.poisson, 1, R.txc[0], R.txc[1]), "tlab_poisson_fft_x");              // p -> tmp2 (complex slab)
        repack(d, R.txc[1], R.pack[0], 1);
    }
    twait(d, pencil_exchange(d, 0, 0, true));
    for (Rank &R : d->rk) {
        ok(tlab_poisson_fft_z(R.poisson, 1, R.pen[0], R.pen[1]), "tlab_poisson_fft_z");
        if (direct) ok(tlab_poisson_direct_ode(R.poisson, TLAB_BCS_NN, R.pen[1], R.pen[1]), "tlab_poisson_direct_ode");   // p^ over f^
        else ok(tlab_poisson_ode(R.poisson, R.pen[1], R.pen[1], R.pen[2]), "tlab_poisson_ode");                           // p^ over f^, dp^ in pen[2]
        ok(tlab_poisson_fft_z(R.poisson, -1, R.pen[1], R.pen[0]), "tlab_poisson_fft_z");
    }
    const int w0 = pencil_exchange(d, 0, 0, false);                                                   // p travels ...
    if (direct) {    // one field on the way back; dp/dy = OPR_Partial_Y(p) on the slab (opr_elliptic.f90:447-449)
        twait(d, w0);
        for (Rank &R : d->rk) {
            if (d->fused_x) x_from_pack(d, R, R.pack[0], R.txc[0], false);
            else {
                repack(d, R.txc[1], R.pack[0], -1);
                ok(tlab_poisson_fft_x(R.poisson, -1, R.txc[1], R.txc[0]), "tlab_poisson_fft_x");
            }
            ok(tlab_opr_partial(2, d->g[1], TLAB_OPR_P1, d->nx, d->ny, d->kmax, 0, R.txc[0], R.txc[2], nullptr), "tlab_opr_partial");
        }
        return;
    }
    for (Rank &R : d->rk) ok(tlab_poisson_fft_z(R.poisson, -1, R.pen[2], R.pen[1]), "tlab_poisson_fft_z");   // ... while dp/dy is transformed
    const int w1 = pencil_exchange(d, 1, 1, false);
    twait(d, w0);
    for (Rank &R : d->rk) {
        if (d->fused_x) { x_from_pack(d, R, R.pack[0], R.txc[0], false); continue; }
        repack(d, R.txc[1], R.pack[0], -1);
        ok(tlab_poisson_fft_x(R.poisson, -1, R.txc[1], R.txc[0]), "tlab_poisson_fft_x");             // p -> tmp1
    }
    twait(d, w1);
    for (Rank &R : d->rk) {
        if (d->fused_x) { x_from_pack(d, R, R.pack[1], R.txc[2], true); continue; }
        repack(d, R.txc[3], R.pack[1], -1);
        ok(tlab_poisson_fft_x(R.poisson, -1, R.txc[3], R.txc[2]), "tlab_poisson_fft_x");             // dp/dy -> tmp3
    }
}

// The same solve with every rank's kx range cut in two halves A, B (plans poisson, poisson_b).  The pack buffer holds all A blocks ahead of all B
// blocks, so each half is one all-to-all of its own.  Same kernels on the same modes: the result is that of the one-piece exchange to the bit.
void poisson_pencil_staged(D *d) {
    const int P = d->P, L = (int)d->rk.size(), ny = d->ny, kmax = d->kmax, nzt = d->nzt;
    std::vector<int> nxb((size_t)P);
    for (int p = 0; p < P; ++p) nxb[p] = d->nxl[p] - d->nxa[p];
    auto pen = [&](Rank &R, int i, int h) { return h == 0 ? R.pen[i] : R.pen[i] + 2LL * d->nxa[R.r] * ny * nzt; };
    auto pack = [&](Rank &R, int i, int h) { return h == 0 ? R.pack[i] : R.pack[i] + d->st_split; };
    auto exchange = [&](int i_pen, int i_pack, int h, bool forward) {
        const std::vector<int> &w = h == 0 ? d->nxa : nxb;
        std::vector<double *> send(L), recv(L);
        std::vector<long long> scnt((size_t)L * P), rcnt((size_t)L * P);
        for (int l = 0; l < L; ++l) {
            Rank &R = d->rk[l];
            for (int p = 0; p < P; ++p) {
                const long long slab_cnt = 2LL * w[p] * ny * kmax, pen_cnt = 2LL * w[R.r] * ny * kmax;
                scnt[(size_t)l * P + p] = forward ? slab_cnt : pen_cnt;
                rcnt[(size_t)l * P + p] = forward ? pen_cnt : slab_cnt;
            }
            send[l] = forward ? pack(R, i_pack, h) : pen(R, i_pen, h);
            recv[l] = forward ? pen(R, i_pen, h) : pack(R, i_pack, h);
        }
        return a2a(d, send, scnt, recv, rcnt);
    };
    for (Rank &R : d->rk) {
        ok(tlab_poisson_set_wall_planes(R.poisson, R.txc[0], R.hb, R.ht), "tlab_poisson_set_wall_planes");
        if (d->fused_x) { x_to_pack(d, R, R.txc[0], R.pack[0]); continue; }
        ok(tlab_poisson_fft_x(R.poisson, 1, R.txc[0], R.txc[1]), "tlab_poisson_fft_x");
        ok(tlab_pencil_repack_blocks(R.txc[1], R.pack[0], d->nxh, ny, kmax, 2 * P, d->st_start.data(), d->st_base.data(), 1), "tlab_pencil_repack_blocks");
    }
    int fwd[2], back[2][2];
    for (int h = 0; h < 2; ++h) fwd[h] = exchange(0, 0, h, true);
    for (int h = 0; h < 2; ++h) {
        twait(d, fwd[h]);
        for (Rank &R : d->rk) {
            tlab_poisson_plan_t pl = h == 0 ? R.poisson : R.poisson_b;
            double *b0 = pen(R, 0, h), *b1 = pen(R, 1, h), *b2 = pen(R, 2, h);
            ok(tlab_poisson_fft_z(pl, 1, b0, b1), "tlab_poisson_fft_z");
            ok(tlab_poisson_ode(pl, b1, b1, b2), "tlab_poisson_ode");
            ok(tlab_poisson_fft_z(pl, -1, b1, b0), "tlab_poisson_fft_z");
        }
        back[h][0] = exchange(0, 0, h, false);          // p^ of this half travels (its forward block of pack[0] has been consumed) ...
        for (Rank &R : d->rk)
            ok(tlab_poisson_fft_z(h == 0 ? R.poisson : R.poisson_b, -1, pen(R, 2, h), pen(R, 1, h)), "tlab_poisson_fft_z");   // ... while dp^/dy is transformed
        back[h][1] = exchange(1, 1, h, false);
    }
    const int slab_of[2] = {1, 3}, out_of[2] = {0, 2};  // p -> tmp1, dp/dy -> tmp3
    for (int i = 0; i < 2; ++i) {
        twait(d, back[0][i]);
        twait(d, back[1][i]);
        for (Rank &R : d->rk) {
            if (d->fused_x) { x_from_pack(d, R, R.pack[i], R.txc[out_of[i]], i == 1); continue; }
            ok(tlab_pencil_repack_blocks(R.txc[slab_of[i]], R.pack[i], d->nxh, ny, kmax, 2 * P, d->st_start.data(), d->st_base.data(), -1),
               "tlab_pencil_repack_blocks");
            ok(tlab_poisson_fft_x(R.poisson, -1, R.txc[slab_of[i]], R.txc[out_of[i]]), "tlab_poisson_fft_x");
        }
    }
}

// The z-slab kernels and the pencil pressure step know neither the anelastic density weights nor the dealiasing filters of the operator state
// (tlab_opr_burgers_set_anelastic / _set_dealiasing act on the single-domain operators only): a decomposed run with either would integrate other
// equations than the same tlab.ini on one rank -- refused, at creation and again at every right-hand side (the state may be set later).
void refuse_unsupported_state(const char *who) {
    if (tlab_internal_anelastic())
        throw Fail(TLAB_EUNSUPPORTED, std::string(who) + ": the anelastic formulation is not built into the z-slab driver (single-domain driver only)");
    if (tlab_internal_dealiasing())
        throw Fail(TLAB_EUNSUPPORTED, std::string(who) + ": dealiasing filters are not built into the z-slab driver (single-domain driver only)");
}

tlab_dns_t dns_handle(D *d, Rank &R);

void poisson_pencil(D *d) {
    if (d->stages == 2) poisson_pencil_staged(d);
    else poisson_pencil_single(d);
}

// Same terms as rhs.cpp / rhs_global_incompressible_1.f90:98-398; the z-terms are added last in every equation (the reference's order differs in
// the third equation: rounding only).  tail: fold the RK update into the last pass of every field (TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT).
void rhs_halo(D *d, double dte, bool tail, double tdte, double kco, int scale) {
    need_bound(d);
    refuse_unsupported_state("tlab_slab_dns_rhs");
    const int nx = d->nx, ny = d->ny, kmax = d->kmax, ns = d->nscal;
    const long long n = d->n;
    const int S_HQ3 = 3 + ns, S_P = 4 + ns;            // halo slots (Rank::lo, hi): 0..2 q, 3..2+ns s, then hq(3) and tmp1
    const bool fresh = d->fresh;       // start of a Runge-Kutta step: hq = hs = 0 (time.f90:212-216) -> the first term of every equation overwrites
    d->fresh = false;
    // dynamic surface model: keep the old tendency of the scalar at the boundary (rhs_global_incompressible_1.f90:77-87); zero at the start of a step,
    // when the tendencies COUNT as zero, and on the sides without a surface model (as rhs.cpp)
    auto surface = [&](int is) { return !d->sfc_jmin.empty() && (d->sfc_jmin[is] == 1 || d->sfc_jmax[is] == 1); };
    for (int is = 0; is < ns; ++is) {
        if (!surface(is)) continue;
        const size_t pbytes = (size_t)nx * kmax * sizeof(double);
        for (Rank &R : d->rk) {
            if (!fresh) hk(tlab::launch_get_wall_planes(R.hs[is], R.sref_b[is], R.sref_t[is], nx, ny, kmax, tlab_current_stream()), "k_get_wall_planes");
            if (fresh || d->sfc_jmin[is] != 1) hk(hipMemsetAsync(R.sref_b[is], 0, pbytes, tlab_current_stream()), "hipMemsetAsync");
            if (fresh || d->sfc_jmax[is] != 1) hk(hipMemsetAsync(R.sref_t[is], 0, pbytes, tlab_current_stream()), "hipMemsetAsync");
        }
    }
    const double idte = d->remove_divergence ? 1.0 / dte : 0.0;      // hq + 0 q is hq bit for bit: the same kernels serve the else-branch (as rhs.cpp)
    // The reference's DEFAULT walls (free-slip u, w; Neumann scalars: boundary_bcs.f90:102-190) without a derivative pass per Neumann field, as in the
    // single-domain driver (rhs.cpp): the wall tendency of BOUNDARY_BCS_NEUMANN_Y is a linear functional of the y line with weights that decay
    // like the coupling of the compact system, so a field is finished with zero wall tendencies by the kernel that holds its last term (Dirichlet
    // treatment), the weighted sums over the STORED tendencies next to the walls give the wall planes (k_wall_weighted) and k_wall_fix sets them.  All of
    // it is local in y: no exchange.  TLAB_NEUMANN_PLANES=0 / TLAB_SLAB_FUSED_X=0 keep the derivative pass (the Python driver's sequence).
    auto ibc_of = [](int tmin, int tmax) { return (tmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (tmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0); };
    const char *npe = std::getenv("TLAB_NEUMANN_PLANES");
    bool planes_route = tail && d->fused_x && !(npe && std::atoi(npe) == 0);
    auto weights_of = [&](Rank &R, int ibc, const double *&w, int &K) { return tlab_internal_dns_neumann_weights(dns_handle(d, R), ibc, &w, &K) == 1; };
    if (planes_route) {      // every Neumann variant in use needs its weights on every local rank
        for (Rank &R : d->rk) {
            const double *w;
            int K;
            for (int i = 0; i < 3; i += 2)
                if (int ibc = ibc_of(d->flow_jmin[i], d->flow_jmax[i])) planes_route = planes_route && weights_of(R, ibc, w, K);
            for (int i = 0; i < ns; ++i)
                if (int ibc = ibc_of(d->scal_jmin[i], d->scal_jmax[i])) planes_route = planes_route && weights_of(R, ibc, w, K);
        }
    }
    // wall planes of a field that was finished with zero wall tendencies
    auto wall_fix = [&](Rank &R, double *q, double *h, int ibc) {
        const double *w;
        int K;
        if (!weights_of(R, ibc, w, K)) throw Fail(TLAB_EINVAL, "internal: wall-plane weights vanished");
        hk(tlab::launch_wall_weighted(h, nullptr, w, w + K, K, R.hb, R.ht, nullptr, nullptr, nx, ny, kmax, tlab_current_stream()), "k_wall_weighted");
        hk(tlab::launch_wall_fix(q, h, (ibc & 1) ? R.hb : nullptr, (ibc & 2) ? R.ht : nullptr, tdte, kco, scale, nx, ny, kmax, tlab_current_stream()), "k_wall_fix");
    };
    // scalars are finished by the z pass itself: Dirichlet walls, or Neumann ones on the wall-plane route (TLAB_SLAB_FUSED_X=0: separate update, the
    // Python driver's sequence)
    std::vector<int> zfin((size_t)(3 + ns), 0);
    for (int i = 0; i < ns; ++i)
        zfin[3 + i] = tail && d->fused_x && !surface(i) && (ibc_of(d->scal_jmin[i], d->scal_jmax[i]) == 0 || planes_route);
    // ---- diffusion + advection (:98-162) and the pressure forcing div(hq + q/dte) (:188-260) ----
    // (Measured in round 4 and dropped: the x and y terms of the forcing inside the Burgers launches that add the last term of u resp. v, as rhs.cpp
    // does on one device.  It needs u to end with its x term and v with its y term, i.e. the x and y launches split in two; on slabs of 64 planes the
    // extra velocity reads and the smaller launches cost what the two saved passes return: 28.17 against 28.04 ms for 8 loopback ranks at 512^3.)
    int w = halo_start(d, 3 + ns, [&](Rank &R, int i) { return Slot{i < 3 ? R.q[i] : R.s[i - 3], i}; });
    for (Rank &R : d->rk) badd_all(d, R, 1, fresh);
    twait(d, w);
    for (Rank &R : d->rk) zburgers_all(d, R, 1);
    w = msg_start(d, 2 * (3 + ns));
    for (Rank &R : d->rk) badd_all(d, R, 2, false);
    twait(d, w);
    for (Rank &R : d->rk) zburgers_all(d, R, 2, &zfin, tdte, kco, scale);
    w = halo_start(d, 1, [&](Rank &R, int) { return Slot{R.hq[2], S_HQ3}; });          // w's halo planes are still valid
    for (Rank &R : d->rk) padd(d, R, 2, R.hq[1], R.q[1], idte, R.txc[0], 0);
    for (Rank &R : d->rk) padd(d, R, 1, R.hq[0], R.q[0], idte, R.txc[0], 1);
    twait(d, w);
    for (Rank &R : d->rk) zpartial(d, R, 1, R.hq[2], S_HQ3, R.q[2], 2, idte, nullptr, 0);
    w = msg_start(d, 1);
    for (Rank &R : d->rk) ok(tlab_pw_get_wall_planes(R.hq[1], R.hb, R.ht, nx, ny, kmax), "tlab_pw_get_wall_planes");
    twait(d, w);
    for (Rank &R : d->rk) zpartial(d, R, 2, R.hq[2], S_HQ3, R.q[2], 2, idte, R.txc[0], 1);
    // ---- pressure (:284) and its gradient (:319-320) ----
    auto dirichlet = [](int t) { return t == TLAB_DNS_BCS_DIRICHLET; };
    bool vel_dirichlet = true, scal_dirichlet = true;
    for (int i = 0; i < 3; ++i) vel_dirichlet = vel_dirichlet && dirichlet(d->flow_jmin[i]) && dirichlet(d->flow_jmax[i]);
    for (int i = 0; i < ns; ++i) scal_dirichlet = scal_dirichlet && dirichlet(d->scal_jmin[i]) && dirichlet(d->scal_jmax[i]);
    const bool grad_final = tail && (vel_dirichlet || planes_route);      // (v is always Dirichlet: tlab_slab_dns_set_bcs)
    const bool v_final = grad_final && d->fused_x && !d->gy_elliptic;      // v is finished by the inverse x-transform of dp^/dy
    d->vf.armed = v_final; d->vf.dte = tdte; d->vf.kco = kco; d->vf.scale = scale;
    poisson_pencil(d);
    d->vf.armed = false;
    // BOUNDARY_BCS_SURFACE_Y (boundary_bcs.f90:478-546) on slabs: d/dy of the scalar is local; AVG1V2D of its plane j = 1 (at BOTH ends, as the reference
    // has it) is the average over all ranks -- local averages to the host, one all-reduce (sum) over the z communicator, the global average back to the
    // device; the flux anomaly then goes into BcsScal%ref rank by rank.  A synchronising step: the model is a correctness feature of examples/Case88.
    std::vector<double> sfc_avg_host;      // [scalar]: the global plane average of d s / dy, computed once per substep for every surface scalar
    auto surface_averages = [&]() {
        sfc_avg_host.assign((size_t)ns, 0.0);
        bool any = false;
        for (int is = 0; is < ns; ++is) any = any || surface(is);
        if (!any) return;
        const int L = (int)d->rk.size();
        std::vector<double> v((size_t)L * ns, 0.0);
        for (int is = 0; is < ns; ++is) {
            if (!surface(is)) continue;
            for (int l = 0; l < L; ++l) {
                Rank &R = d->rk[l];
                ok(tlab_opr_partial(2, d->g[1], TLAB_OPR_P1, nx, ny, kmax, 0, R.s[is], R.txc[4], nullptr), "OPR_Partial_Y (surface flux)");      // boundary_bcs.f90:508
                hk(tlab::launch_plane_avg(R.txc[4], 0, nx, ny, kmax, R.sfc_avg, tlab_current_stream()), "k_plane_sum");
                hk(hipMemcpyAsync(&v[(size_t)l * ns + is], R.sfc_avg, sizeof(double), hipMemcpyDeviceToHost, tlab_current_stream()), "hipMemcpyAsync");
            }
        }
        hk(hipStreamSynchronize(tlab_current_stream()), "hipStreamSynchronize");
        tck(d->tr.allreduce(d->tr.ctx, v.data(), ns, 2), "allreduce (sum)");            // sum of the ranks' plane averages
        for (int is = 0; is < ns; ++is) sfc_avg_host[is] = v[is] / (double)d->P;
    };
    auto surface_flux_local = [&](Rank &R, int is) {
        const double diff = d->visc / d->schmidt[is];
        ok(tlab_opr_partial(2, d->g[1], TLAB_OPR_P1, nx, ny, kmax, 0, R.s[is], R.txc[4], nullptr), "OPR_Partial_Y (surface flux)");
        if (d->sfc_jmin[is] == 1)
            hk(tlab::launch_surface_flux_avg(R.sref_b[is], R.txc[4], 0, 1.0, diff, d->cpl_jmin[is], sfc_avg_host[is], nx, ny, kmax, tlab_current_stream()), "k_surface_flux");
        if (d->sfc_jmax[is] == 1)
            hk(tlab::launch_surface_flux_avg(R.sref_t[is], R.txc[4], ny - 1, -1.0, diff, d->cpl_jmax[is], sfc_avg_host[is], nx, ny, kmax, tlab_current_stream()), "k_surface_flux");
    };
    surface_averages();
    // ---- hq -= grad p, boundary conditions (:348-398) [+ RK update] ----
    auto finish = [&](Rank &R) {
        struct Fd {
            double *q, *h, *g;
            int tmin, tmax, is;
        };
        std::vector<Fd> F;
        for (int i = 0; i < 3; ++i) F.push_back({R.q[i], R.hq[i], R.txc[1 + i], d->flow_jmin[i], d->flow_jmax[i], -1});
        for (int i = 0; i < ns; ++i)
            if (!zfin[3 + i]) F.push_back({R.s[i], R.hs[i], nullptr, d->scal_jmin[i], d->scal_jmax[i], i});
        if (grad_final) F.erase(F.begin() + 2), F.erase(F.begin());          // v and the scalars; u, w are done (+ their wall planes below)
        if (v_final) F.erase(F.begin());                                     // the scalars
        if (!grad_final && (!vel_dirichlet || !tail)) {
            ok(tlab_pw_sub3(R.hq[0], R.hq[1], R.hq[2], R.txc[1], R.txc[2], R.txc[3], n), "tlab_pw_sub3");
            for (Fd &f : F) f.g = nullptr;
        }
        for (const Fd &f : F) {
            const int ibc = (f.tmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (f.tmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0);
            if (ibc)       // needs the finished tendency (g is null here)
                ok(tlab_boundary_bcs_neumann_y(d->g[1], ibc, nx, ny, kmax, f.h, R.hb, R.ht, R.txc[0]), "tlab_boundary_bcs_neumann_y");
            const double *pb = (ibc & 1) ? R.hb : nullptr, *pt = (ibc & 2) ? R.ht : nullptr;
            if (f.is >= 0 && surface(f.is)) {      // BcsScal%ref: the Neumann value replaces the kept tendency, the flux anomaly was added by surface_fluxes
                const size_t pbytes = (size_t)nx * kmax * sizeof(double);
                if (pb) hk(hipMemcpyAsync(R.sref_b[f.is], pb, pbytes, hipMemcpyDeviceToDevice, tlab_current_stream()), "hipMemcpyAsync");
                if (pt) hk(hipMemcpyAsync(R.sref_t[f.is], pt, pbytes, hipMemcpyDeviceToDevice, tlab_current_stream()), "hipMemcpyAsync");
                surface_flux_local(R, f.is);
                pb = R.sref_b[f.is]; pt = R.sref_t[f.is];
            }
            if (!tail) ok(tlab_pw_set_wall_planes(f.h, pb, pt, nx, ny, kmax), "tlab_pw_set_wall_planes");
            else ok(tlab_pw_final_update(f.q, f.h, f.g, pb, pt, tdte, kco, scale, nx, ny, kmax), "tlab_pw_final_update");
        }
    };
    w = halo_start(d, 1, [&](Rank &R, int) { return Slot{R.txc[0], S_P}; });
    if (grad_final) {   // u and w are finished by the gradient kernels themselves (no gradient array)
        for (Rank &R : d->rk) {
            ok(tlab_opr_gradient_final(1, d->g[0], nx, ny, kmax, R.txc[0], R.q[0], R.hq[0], tdte, kco, scale, R.txc[1]), "tlab_opr_gradient_final");
            if (int ibc = ibc_of(d->flow_jmin[0], d->flow_jmax[0])) wall_fix(R, R.q[0], R.hq[0], ibc);
        }
    } else {
        for (Rank &R : d->rk) padd(d, R, 1, R.txc[0], nullptr, 0.0, R.txc[1], 0);
    }
    twait(d, w);
    for (Rank &R : d->rk) zpartial(d, R, 1, R.txc[0], S_P, nullptr, 0, 0.0, nullptr, 0);
    w = msg_start(d, 1);
    // v and the scalars do not wait for dp/dz: their update runs while the interface values travel (not with Neumann scalars, whose boundary
    // routine takes tmp1 = p as scratch)
    bool needs_bcs_routine = false;      // a field left for `finish` with a Neumann wall: BOUNDARY_BCS_NEUMANN_Y, which takes tmp1 = p as scratch
    for (int i = 0; i < ns; ++i) needs_bcs_routine = needs_bcs_routine || (!zfin[3 + i] && ibc_of(d->scal_jmin[i], d->scal_jmax[i]) != 0);
    const bool early_finish = grad_final && (scal_dirichlet || !needs_bcs_routine);
    if (early_finish)
        for (Rank &R : d->rk) finish(R);
    twait(d, w);
    if (grad_final) {
        for (Rank &R : d->rk) {
            const double *ph[2] = {R.lo[S_P], R.hi[S_P]};
            ok(tlab_internal_zslab_gradient_final_z(R.zplan, nx, ny, R.txc[0], ph, R.tail_left, R.head_right, R.q[2], R.hq[2], tdte, kco, scale),
               "tlab_zslab_gradient_final_z");
            if (int ibc = ibc_of(d->flow_jmin[2], d->flow_jmax[2])) wall_fix(R, R.q[2], R.hq[2], ibc);
        }
    } else {
        for (Rank &R : d->rk) zpartial(d, R, 2, R.txc[0], S_P, nullptr, 0, 0.0, R.txc[3], 0);
    }
    if (!early_finish)
        for (Rank &R : d->rk) finish(R);
    // Neumann scalars the z pass finished: their wall planes
    for (int i = 0; i < ns; ++i)
        if (zfin[3 + i])
            if (int ibc = ibc_of(d->scal_jmin[i], d->scal_jmax[i]))
                for (Rank &R : d->rk) wall_fix(R, R.s[i], R.hs[i], ibc);
}

tlab_dns_t dns_handle(D *d, Rank &R) {
    if (!R.dns) {
        const double one = 1.0;
        ok(tlab_dns_create(&R.dns, d->g[0], d->g[1], d->g[2], R.poisson, d->nx, d->ny, d->kmax, d->nscal, d->visc, d->nscal ? d->schmidt.data() : &one),
           "tlab_dns_create");
        ok(tlab_dns_set_slab(R.dns, R.r * d->kmax), "tlab_dns_set_slab");
    }
    return R.dns;
}

template <class F>
int guarded(F f) {
    try {
        if (!tlab_device_ready()) throw Fail(TLAB_EHIP, "tlab_init has not been called (no CPU fallback exists)");
        f();
        return TLAB_OK;
    } catch (const Fail &e) {
        tlab_set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

}  // namespace

extern "C" {

int tlab_slab_transport_loopback(tlab_slab_transport *out, int nranks) {
    if (!out || nranks < 1) { tlab_set_error("tlab_slab_transport_loopback: bad arguments"); return TLAB_EINVAL; }
    out->ctx = new Loopback{nranks};
    out->nranks = nranks; out->nlocal = nranks; out->first = 0;
    out->ring_start = lb_ring; out->alltoallv_start = lb_a2a; out->wait = lb_wait; out->allreduce = lb_allreduce; out->destroy = lb_destroy;
    return TLAB_OK;
}

int tlab_slab_dns_create(tlab_slab_dns_t *out, const tlab_slab_transport *tr, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx, int ny,
                         int nz_total, int nscal, double visc, const double *schmidt, tlab_fdm_plan_t gy_elliptic) {
    return guarded([&] {
        if (!out || !tr || !gx || !gy || !gz || nscal < 0 || (nscal > 0 && !schmidt) || visc <= 0.0) throw Fail(TLAB_EINVAL, "tlab_slab_dns_create: bad arguments");
        if (!tr->ring_start || !tr->alltoallv_start || !tr->wait || !tr->allreduce) throw Fail(TLAB_EINVAL, "tlab_slab_dns_create: incomplete transport");
        const int P = tr->nranks;
        if (P < 1) throw Fail(TLAB_EINVAL, "tlab_slab_dns_create: nranks < 1");       // (one rank: its own ring neighbour; tests of a transport on one GPU)
        if (tr->nlocal < 1 || tr->first < 0 || tr->first + tr->nlocal > P) throw Fail(TLAB_EINVAL, "tlab_slab_dns_create: local ranks outside [0, nranks)");
        if (nz_total % P) throw Fail(TLAB_EINVAL, "nz must be divisible by the number of z slabs");
        // every cheap refusal BEFORE anything is allocated (on failure the caller keeps ownership of transport->ctx: see include/tlab_amd.h)
        if (P > 16) throw Fail(TLAB_EUNSUPPORTED, "tlab_slab_dns_create: at most 16 z slabs (one node)");
        if ((nx / 2 + 1) / P < 1) throw Fail(TLAB_EINVAL, "fewer kx modes than ranks");
        refuse_unsupported_state("tlab_slab_dns_create");
        auto d = std::make_unique<tlab_slab_dns>();
        d->g[0] = gx; d->g[1] = gy; d->g[2] = gz; d->gy_elliptic = gy_elliptic;
        d->P = P; d->nx = nx; d->ny = ny; d->nzt = nz_total; d->kmax = nz_total / P; d->nxh = nx / 2 + 1; d->nscal = nscal; d->visc = visc;
        d->npage = (long long)nx * ny; d->n = d->npage * d->kmax;
        if (nscal) d->schmidt.assign(schmidt, schmidt + nscal);
        d->scal_jmin.assign(nscal, TLAB_DNS_BCS_DIRICHLET);
        d->scal_jmax.assign(nscal, TLAB_DNS_BCS_DIRICHLET);
        // kx ranges of the pencils: [ioff[r], ioff[r] + nxl[r])
        const int base = d->nxh / P, rem = d->nxh % P;
        for (int r = 0; r < P; ++r) {
            d->nxl.push_back(base + (r < rem ? 1 : 0));
            d->ioff.push_back(r * base + std::min(r, rem));
            d->nxa.push_back((d->nxl[r] + 1) / 2);
        }
        const char *env = std::getenv("TLAB_PENCIL_STAGES");
        d->stages = (!gy_elliptic && base >= 2 && 2 * P <= 16 && !(env && std::strcmp(env, "1") == 0)) ? 2 : 1;
        if (d->stages == 2) {   // block 2p / 2p+1 = half A / B of rank p: all A blocks (by rank) ahead of all B blocks
            long long offa = 0, offb = 0;
            for (int p = 0; p < P; ++p) offb += (long long)d->nxa[p] * ny * d->kmax;
            d->st_split = 2 * offb;
            for (int p = 0; p < P; ++p) {
                d->st_start.push_back(d->ioff[p]);
                d->st_start.push_back(d->ioff[p] + d->nxa[p]);
                d->st_base.push_back(offa);
                d->st_base.push_back(offb);
                offa += (long long)d->nxa[p] * ny * d->kmax;
                offb += (long long)(d->nxl[p] - d->nxa[p]) * ny * d->kmax;
            }
        }
        {
            long long off = 0;
            for (int p = 0; p < P; ++p) { d->sg_start.push_back(d->ioff[p]); d->sg_base.push_back(off); off += (long long)d->nxl[p] * ny * d->kmax; }
        }
        d->rk.resize(tr->nlocal);
        const int nmsg = 2 * (3 + nscal);
        for (int l = 0; l < tr->nlocal; ++l) {
            Rank &R = d->rk[l];
            R.r = tr->first + l;
            ok(tlab_zslab_plan_create(&R.zplan, gz, d->kmax, R.r * d->kmax, 0), "tlab_zslab_plan_create");
            if (gy_elliptic)
                ok(tlab_poisson_plan_create_direct_decomposed(&R.poisson, gx, gy, gz, nx, ny, d->kmax, nz_total, 1, d->ioff[R.r], d->nxl[R.r], gy_elliptic),
                   "tlab_poisson_plan_create_direct_decomposed");
            else if (d->stages == 2) {
                ok(tlab_poisson_plan_create_pencil(&R.poisson, gx, gy, gz, nx, ny, d->kmax, nz_total, d->ioff[R.r], d->nxa[R.r]), "tlab_poisson_plan_create_pencil");
                ok(tlab_poisson_plan_create_pencil(&R.poisson_b, gx, gy, gz, nx, ny, d->kmax, nz_total, d->ioff[R.r] + d->nxa[R.r], d->nxl[R.r] - d->nxa[R.r]),
                   "tlab_poisson_plan_create_pencil");
            } else
                ok(tlab_poisson_plan_create_pencil(&R.poisson, gx, gy, gz, nx, ny, d->kmax, nz_total, d->ioff[R.r], d->nxl[R.r]), "tlab_poisson_plan_create_pencil");
            R.hb = dalloc((size_t)nx * d->kmax);
            R.ht = dalloc((size_t)nx * d->kmax);
            for (double **p : {&R.head, &R.tail, &R.head_right, &R.tail_left}) *p = dalloc((size_t)nmsg * d->npage);
            for (int i = 0; i < 3; ++i) R.pen[i] = dalloc((size_t)2 * d->nxl[R.r] * ny * nz_total);
            for (int i = 0; i < 2; ++i) R.pack[i] = dalloc((size_t)2 * d->nxh * ny * d->kmax);
            const int nslots = 3 + nscal + 2;
            const size_t Hn = (size_t)HALO * d->npage;
            R.halo = dalloc((size_t)nslots * 2 * Hn);
            for (int i = 0; i < nslots; ++i) { R.lo.push_back(R.halo + (size_t)(2 * i) * Hn); R.hi.push_back(R.halo + (size_t)(2 * i + 1) * Hn); }
        }
        const char *fx = std::getenv("TLAB_SLAB_FUSED_X");
        d->fused_x = !(fx && std::strcmp(fx, "0") == 0);
        for (Rank &R : d->rk) d->fused_x = d->fused_x && tlab_internal_poisson_has_own_x(R.poisson);
        d->tr = *tr;         // from here on the driver owns the transport's context
        *out = d.release();
    });
}

int tlab_slab_dns_destroy(tlab_slab_dns_t d) {
    (void)tlab_internal_deferred_flush();
    delete d;
    return TLAB_OK;
}

int tlab_slab_dns_bind(tlab_slab_dns_t d, int l, double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc) {
    return guarded([&] {
        if (!d || l < 0 || l >= (int)d->rk.size() || !q || !hq || !txc || (d->nscal > 0 && (!s || !hs))) throw Fail(TLAB_EINVAL, "tlab_slab_dns_bind: bad arguments");
        Rank &R = d->rk[l];
        R.q.assign(q, q + 3); R.hq.assign(hq, hq + 3); R.txc.assign(txc, txc + 9);
        R.s.assign(s, s + d->nscal); R.hs.assign(hs, hs + d->nscal);
        for (const std::vector<double *> *v : {&R.q, &R.s, &R.hq, &R.hs, &R.txc})
            for (double *p : *v)
                if (!p) throw Fail(TLAB_EINVAL, "tlab_slab_dns_bind: null array");
        R.bound = true;
    });
}

long long tlab_slab_dns_info(tlab_slab_dns_t d, int what) {
    if (!d) return TLAB_EINVAL;
    switch (what) {
        case 0: return d->kmax;
        case 1: return d->P;
        case 2: return (long long)d->rk.size();
        case 3: return HALO * d->npage;
        case 4: return d->stages;
        case 5: return d->rk.empty() ? 0 : d->rk[0].r;
        case 6: return d->fused_x ? 1 : 0;
    }
    return TLAB_EINVAL;
}

int tlab_slab_dns_set_bcs(tlab_slab_dns_t d, const int *flow_jmin, const int *flow_jmax, const int *scal_jmin, const int *scal_jmax) {
    return guarded([&] {
        if (!d || !flow_jmin || !flow_jmax || (d->nscal > 0 && (!scal_jmin || !scal_jmax))) throw Fail(TLAB_EINVAL, "tlab_slab_dns_set_bcs: bad arguments");
        auto valid = [](int t) { return t == TLAB_DNS_BCS_DIRICHLET || t == TLAB_DNS_BCS_NEUMANN; };
        for (int i = 0; i < 3; ++i)
            if (!valid(flow_jmin[i]) || !valid(flow_jmax[i])) throw Fail(TLAB_EINVAL, "tlab_slab_dns_set_bcs: type must be DNS_BCS_DIRICHLET or DNS_BCS_NEUMANN");
        for (int i = 0; i < d->nscal; ++i)
            if (!valid(scal_jmin[i]) || !valid(scal_jmax[i])) throw Fail(TLAB_EINVAL, "tlab_slab_dns_set_bcs: type must be DNS_BCS_DIRICHLET or DNS_BCS_NEUMANN");
        if (flow_jmin[1] != TLAB_DNS_BCS_DIRICHLET || flow_jmax[1] != TLAB_DNS_BCS_DIRICHLET)
            throw Fail(TLAB_EUNSUPPORTED, "tlab_slab_dns_set_bcs: the wall-normal velocity must be Dirichlet (impermeable walls; the pressure BCs assume v = 0)");
        for (int i = 0; i < 3; ++i) { d->flow_jmin[i] = flow_jmin[i]; d->flow_jmax[i] = flow_jmax[i]; }
        for (int i = 0; i < d->nscal; ++i) { d->scal_jmin[i] = scal_jmin[i]; d->scal_jmax[i] = scal_jmax[i]; }
    });
}

int tlab_slab_dns_set_remove_divergence(tlab_slab_dns_t d, int on) {
    if (!d) return TLAB_EINVAL;
    d->remove_divergence = on != 0;
    return TLAB_OK;
}

int tlab_slab_dns_set_surface_bcs(tlab_slab_dns_t d, const int *sfc_jmin, const int *sfc_jmax, const double *cpl_jmin, const double *cpl_jmax) {
    return guarded([&] {
        if (!d || (d->nscal > 0 && (!sfc_jmin || !sfc_jmax || !cpl_jmin || !cpl_jmax))) throw Fail(TLAB_EINVAL, "tlab_slab_dns_set_surface_bcs: bad arguments");
        for (int i = 0; i < d->nscal; ++i)
            if ((sfc_jmin[i] != 0 && sfc_jmin[i] != 1) || (sfc_jmax[i] != 0 && sfc_jmax[i] != 1)) throw Fail(TLAB_EINVAL, "SfcType: 0 static or 1 linear");
        d->sfc_jmin.assign(sfc_jmin, sfc_jmin + d->nscal); d->sfc_jmax.assign(sfc_jmax, sfc_jmax + d->nscal);
        d->cpl_jmin.assign(cpl_jmin, cpl_jmin + d->nscal); d->cpl_jmax.assign(cpl_jmax, cpl_jmax + d->nscal);
        for (Rank &R : d->rk) {
            R.sref_b.resize(d->nscal, nullptr); R.sref_t.resize(d->nscal, nullptr);
            for (int i = 0; i < d->nscal; ++i)
                if ((sfc_jmin[i] == 1 || sfc_jmax[i] == 1) && !R.sref_b[i]) {
                    R.sref_b[i] = dalloc((size_t)d->nx * d->kmax);
                    R.sref_t[i] = dalloc((size_t)d->nx * d->kmax);
                    hk(hipMemsetAsync(R.sref_b[i], 0, (size_t)d->nx * d->kmax * sizeof(double), tlab_current_stream()), "hipMemsetAsync");
                    hk(hipMemsetAsync(R.sref_t[i], 0, (size_t)d->nx * d->kmax * sizeof(double), tlab_current_stream()), "hipMemsetAsync");
                }
            if (!R.sfc_avg) R.sfc_avg = dalloc(1);
        }
    });
}

int tlab_slab_dns_begin_step(tlab_slab_dns_t d) {
    (void)tlab_internal_deferred_flush();
    if (!d) return TLAB_EINVAL;
    d->fresh = true;
    return TLAB_OK;
}

int tlab_slab_dns_rhs(tlab_slab_dns_t d, double dte) {
    return guarded([&] {
        if (!d || !(dte > 0.0)) throw Fail(TLAB_EINVAL, "tlab_slab_dns_rhs: bad arguments");
        rhs_halo(d, dte, false, 0.0, 1.0, 0);
    });
}

int tlab_slab_dns_substep(tlab_slab_dns_t d, double dte, double kco, int scale_tendencies) {
    return guarded([&] {
        if (!d || !(dte > 0.0)) throw Fail(TLAB_EINVAL, "tlab_slab_dns_substep: bad arguments");
        rhs_halo(d, dte, true, dte, kco, scale_tendencies);
    });
}

int tlab_slab_dns_time_courant(tlab_slab_dns_t d, double cfla, double cfld, double *pmax, double *dtime) {
    return guarded([&] {
        if (!d || !pmax) throw Fail(TLAB_EINVAL, "tlab_slab_dns_time_courant: bad arguments");
        need_bound(d);
        const int L = (int)d->rk.size();
        std::vector<double> v((size_t)2 * L);
        for (int l = 0; l < L; ++l) ok(tlab_time_courant(dns_handle(d, d->rk[l]), d->rk[l].q.data(), cfla, cfld, &v[(size_t)2 * l], nullptr), "tlab_time_courant");
        tck(d->tr.allreduce(d->tr.ctx, v.data(), 2, 0), "allreduce");            // MPI_MAX, time.f90:522
        pmax[0] = v[0]; pmax[1] = v[1];
        if (dtime) {
            const double dtc = pmax[0] > 0.0 ? cfla / pmax[0] : 1.0e300, dtd = pmax[1] > 0.0 ? cfld / pmax[1] : 1.0e300;
            *dtime = cfla > 0.0 ? std::min(dtc, dtd) : 0.0;
        }
    });
}

int tlab_slab_dns_dilatation_bounds(tlab_slab_dns_t d, double *dil_min, double *dil_max) {
    return guarded([&] {
        if (!d || !dil_min || !dil_max) throw Fail(TLAB_EINVAL, "tlab_slab_dns_dilatation_bounds: bad arguments");
        need_bound(d);
        const int nx = d->nx, ny = d->ny, kmax = d->kmax, L = (int)d->rk.size();
        // div(q) with the z-derivative by the slab route of the RHS (FI_INVARIANT_P = -div, fi_vectorcalculus.f90:111-141)
        int w = halo_start(d, 1, [&](Rank &R, int) { return Slot{R.q[2], 2}; });
        for (Rank &R : d->rk) {
            ok(tlab_opr_partial(1, d->g[0], TLAB_OPR_P1, nx, ny, kmax, 0, R.q[0], R.txc[0], nullptr), "tlab_opr_partial");
            padd(d, R, 2, R.q[1], nullptr, 0.0, R.txc[0], 1);
        }
        twait(d, w);
        for (Rank &R : d->rk) zpartial(d, R, 1, R.q[2], 2, nullptr, 0, 0.0, nullptr, 0);
        w = msg_start(d, 1);
        twait(d, w);
        for (Rank &R : d->rk) zpartial(d, R, 2, R.q[2], 2, nullptr, 0, 0.0, R.txc[0], 1);
        std::vector<double> mn((size_t)L), mx((size_t)L);
        for (int l = 0; l < L; ++l) ok(tlab_minmax(dns_handle(d, d->rk[l]), d->rk[l].txc[0], nx, ny, kmax, &mn[l], &mx[l]), "tlab_minmax");
        tck(d->tr.allreduce(d->tr.ctx, mn.data(), 1, 1), "allreduce");
        tck(d->tr.allreduce(d->tr.ctx, mx.data(), 1, 0), "allreduce");
        *dil_min = mn[0];
        *dil_max = mx[0];
    });
}

}  // extern "C"

// deferred.cpp: the arrays of the ONE local rank of a Fortran / MPI host (several local ranks -- loopback runs -- have no single DAXPY partner)
bool tlab_internal_slab_bound(tlab_slab_dns_t d, double *const **q, double *const **s, double *const **hq, double *const **hs, int *nscal, long long *n) {
    if (!d || d->rk.size() != 1 || !d->rk[0].bound) return false;
    *q = d->rk[0].q.data(); *s = d->rk[0].s.data(); *hq = d->rk[0].hq.data(); *hs = d->rk[0].hs.data();
    *nscal = d->nscal; *n = d->n;
    return true;
}
