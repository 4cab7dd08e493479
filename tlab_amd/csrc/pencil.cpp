// tlab_pencil_dns_* : RHS_GLOBAL_INCOMPRESSIBLE_1 + TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT on ims_npro_i x ims_npro_k blocks (x/z pencils), behind the C ABI.
//
// The reference's own scheme for a 2-D decomposition (base/tlab_mpi_procs.f90:76-94): every x operator goes through an I-transposition inside
// ims_comm_x (operators/opr_partial.f90:66-86, :133-147; physics/opr_burgers.f90:216-262), every z operator through a K-transposition inside
// ims_comm_z (opr_partial.f90:185-253; opr_burgers.f90:386-426), y is never split; the operator sequence and the reuse of the transposed velocities
// are those of tools/dns/rhs_global_incompressible_1.f90:98-375.  The Poisson solver works on the 1 x (npro_i npro_k) z-slabs the I-transposition
// leaves and on kx-pencils over ALL ranks (one all-to-all after the x transform, one back per output; operators/opr_fourier.f90:232-262 does the
// same through its own transpositions).
//
// This is the C++ port of tlab_amd/pencil.py::PencilDns (the Python statement stays as the cross-check: the two are bit-identical on loopback
// ranks, tests/test_gpu_pencil.py) so that a Fortran / MPI host can call it: host cost per rank and substep < 1 ms instead of ~7 ms.
// Exchanges go through a transport struct of four entry points (tlab_pencil_transport): RCCL (libtlab_amd_comm.so: tlab_comm_pencil_transport), the
// single-process loopback (every rank on one device, exchanges = device copies), or the caller's GPU-aware MPI_Alltoallv.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tlab_amd.h"

extern hipStream_t tlab_current_stream();
int tlab_internal_deferred_flush();      // deferred.cpp
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();
extern "C" bool tlab_internal_anelastic();
extern "C" bool tlab_internal_dealiasing();

namespace tlab {
hipError_t launch_copy_blocks(int n, const double *const *src, double *const *dst, const long long *cnt, hipStream_t st);      // pointwise.hip
hipError_t launch_trp_copy(double *S, double *W, long long m, int P, long long c, int to_wire, hipStream_t st);
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
double *dalloc(size_t n) {
    double *p = nullptr;
    hk(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(double)), "hipMalloc");
    hk(hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(double), tlab_current_stream()), "hipMemset");
    return p;
}

// ---- loopback transport: all npro_i x npro_k ranks in this process ----------------------------------------------------------------------------
struct Loopback {
    int npi, npk;
};
// members of communicator `which` of world rank r, in member order (tlab_mpi_procs.f90:76-94: ims_pro_i = mod(ims_pro, npro_i), ims_pro_k = ims_pro / npro_i)
void members(int npi, int npk, int which, int r, std::vector<int> &m) {
    m.clear();
    const int pi = r % npi, pk = r / npi;
    if (which == 0) for (int q = 0; q < npi * npk; ++q) m.push_back(q);
    else if (which == 1) for (int q = 0; q < npi; ++q) m.push_back(pk * npi + q);
    else for (int q = 0; q < npk; ++q) m.push_back(q * npi + pi);
}
int lb_a2a(void *ctx, void *stream, int which, double *const *send, const long long *scount, double *const *recv, const long long *rcount) {
    const Loopback *L = static_cast<Loopback *>(ctx);
    const int P = L->npi * L->npk;
    std::vector<const double *> sp;
    std::vector<double *> dp;
    std::vector<long long> cn;
    std::vector<int> mem, mem2;
    for (int dst = 0; dst < P; ++dst) {
        members(L->npi, L->npk, which, dst, mem);
        const int S = (int)mem.size();
        int me = 0;
        for (int j = 0; j < S; ++j) if (mem[j] == dst) me = j;
        long long ro = 0;
        for (int j = 0; j < S; ++j) {      // block j of dst's receive buffer comes from member j: its block for member `me`
            const int src = mem[j];
            long long so = 0;
            for (int p = 0; p < me; ++p) so += scount[(size_t)src * S + p];
            const long long cnt = scount[(size_t)src * S + me];
            if (cnt != rcount[(size_t)dst * S + j]) return TLAB_EINVAL;
            if (cnt > 0) { sp.push_back(send[src] + so); dp.push_back(recv[dst] + ro); cn.push_back(cnt); }
            ro += cnt;
        }
    }
    return tlab::launch_copy_blocks((int)sp.size(), sp.data(), dp.data(), cn.data(), (hipStream_t)stream) == hipSuccess ? 0 : TLAB_EHIP;
}
int lb_wait(void *, void *, int) { return TLAB_OK; }
int lb_allreduce(void *ctx, double *v, int n, int op) {
    const Loopback *L = static_cast<Loopback *>(ctx);
    const int P = L->npi * L->npk;
    if (op < 0 || op > 2) return TLAB_EINVAL;      // 0 = max, 1 = min, 2 = sum (as tlab_slab_transport, include/tlab_amd.h)
    for (int i = 0; i < n; ++i) {
        double a = v[i];
        for (int r = 1; r < P; ++r) a = op == 0 ? std::max(a, v[r * n + i]) : (op == 1 ? std::min(a, v[r * n + i]) : a + v[r * n + i]);
        for (int r = 0; r < P; ++r) v[r * n + i] = a;
    }
    return TLAB_OK;
}
void lb_destroy(void *ctx) { delete static_cast<Loopback *>(ctx); }

struct Rank {
    int r = 0, pi = 0, pk = 0;
    tlab_poisson_plan_t poisson = nullptr;
    double *hb = nullptr, *ht = nullptr, *rt = nullptr, *u_t = nullptr, *w_t = nullptr, *ta = nullptr, *tb = nullptr, *wire = nullptr;
    // second set for the overlapped schedule (rhs_overlapped): two transposed operators are in flight at a time; tx = one more result field
    double *rt2 = nullptr, *wire2 = nullptr, *wireb[2] = {nullptr, nullptr}, *tx = nullptr;
    double *pen[3] = {nullptr, nullptr, nullptr}, *pack[2] = {nullptr, nullptr};
    std::vector<double *> q, s, hq, hs, txc;
    bool bound = false;
};

}  // namespace

struct tlab_pencil_dns {
    tlab_pencil_transport tr{};
    tlab_fdm_plan_t g[3] = {nullptr, nullptr, nullptr};
    int npi = 1, npk = 1, P = 1, nx = 0, ny = 0, nzt = 0, imax = 0, kmax = 0, kmax2 = 0, nxh = 0, nscal = 0;
    long long n = 0, npage_i = 0, nlx = 0, npage_k = 0, nlz = 0, isize_txc = 0;
    double visc = 0.0;
    std::vector<double> schmidt;
    std::vector<int> nxl, ioff;
    int flow_jmin[3] = {TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET};
    int flow_jmax[3] = {TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET, TLAB_DNS_BCS_DIRICHLET};
    std::vector<int> scal_jmin, scal_jmax;
    bool fresh = false;
    bool overlap = true;              // rhs_overlapped (exchanges started ahead of independent launches) instead of the literal sequence; TLAB_PENCIL_OVERLAP=0
    // tlab_pencil_dns_substep: the velocities are finished by ONE pass each behind the pressure gradient (h -= dp/dx_i, zero wall planes, q += dte h,
    // h *= kco: k_final_update, the arithmetic of k_sub3 + k_set_wall_planes + k_rk_update in the same order) where all their walls are Dirichlet;
    // TLAB_PENCIL_FINAL=0 keeps the three passes
    struct { bool on = false, done = false; double dte = 0.0, kco = 1.0; int scale = 0; } fin;
    bool tracing = false;             // tlab_pencil_dns_trace: the order of exchange starts, launches and waits of the last RHS
    std::string trace;
    long long launches = 0;           // launches issued so far by the overlapped schedule (a wait looks whether any followed its start)
    std::vector<Rank> rk;
    ~tlab_pencil_dns() {
        for (Rank &R : rk) {
            if (R.poisson) (void)tlab_poisson_plan_destroy(R.poisson);
            for (double *p : {R.hb, R.ht, R.rt, R.u_t, R.w_t, R.ta, R.tb, R.wire, R.pen[0], R.pen[1], R.pen[2], R.pack[0], R.pack[1], R.rt2, R.wire2, R.wireb[0], R.wireb[1], R.tx})
                if (p) (void)hipFree(p);
        }
        if (tr.destroy && tr.ctx) tr.destroy(tr.ctx);
    }
};

namespace {
using D = tlab_pencil_dns;

void tck(int t, const char *what) {
    if (t < 0) throw Fail(t, std::string("pencil transport: ") + what + " failed");
}

// equal-block all-to-all inside communicator `which` (1: x, 2: z): buffer sel(R) of every local rank, blk doubles per member
template <class FS, class FR>
void a2a_equal(D *d, int which, long long blk, FS send_of, FR recv_of) {
    const int S = which == 1 ? d->npi : d->npk;
    std::vector<double *> send, recv;
    std::vector<long long> cnt((size_t)d->rk.size() * S, blk);
    for (Rank &R : d->rk) { send.push_back(send_of(R)); recv.push_back(recv_of(R)); }
    const int t = d->tr.alltoallv_start(d->tr.ctx, (void *)tlab_current_stream(), which, send.data(), cnt.data(), recv.data(), cnt.data());
    tck(t, "alltoallv_start");
    tck(d->tr.wait(d->tr.ctx, (void *)tlab_current_stream(), t), "wait");
}
void trp_copy(double *S, double *W, long long m, int P, long long c, int to_wire) {
    hk(tlab::launch_trp_copy(S, W, m, P, c, to_wire, tlab_current_stream()), "k_trp_copy");
}
// TLabMPI_Trp_ExecI_Forward (tlab_mpi_transpose.f90:232-256): a(imax, npage) -> b(imax npro_i, nlines); src / dst: per-rank selectors
template <class FA, class FB>
void trp_i_forward(D *d, FA a_of, FB b_of) {
    const long long blk = d->nlx * d->imax;
    a2a_equal(d, 1, blk, a_of, [](Rank &R) { return R.wire; });                                    // a is blocked by peer as it stands
    for (Rank &R : d->rk) trp_copy(b_of(R), R.wire, d->imax, d->npi, d->nlx, 0);                   // b[line][src][x] = wire[src][line][x]
}
template <class FB, class FA>
void trp_i_backward(D *d, FB b_of, FA a_of) {
    const long long blk = d->nlx * d->imax;
    for (Rank &R : d->rk) trp_copy(b_of(R), R.wire, d->imax, d->npi, d->nlx, 1);
    a2a_equal(d, 1, blk, [](Rank &R) { return R.wire; }, a_of);
}
// TLabMPI_Trp_ExecK_Forward (:301-325): a(npage, kmax) -> b(nlines, kmax npro_k)
template <class FA, class FB>
void trp_k_forward(D *d, FA a_of, FB b_of) {
    const long long blk = d->nlz * d->kmax;
    for (Rank &R : d->rk) trp_copy(a_of(R), R.wire, d->nlz, d->npk, d->kmax, 1);                    // wire[peer][k][line] = a[k][peer nl + line]
    a2a_equal(d, 2, blk, [](Rank &R) { return R.wire; }, b_of);
}
template <class FB, class FA>
void trp_k_backward(D *d, FB b_of, FA a_of) {
    const long long blk = d->nlz * d->kmax;
    a2a_equal(d, 2, blk, b_of, [](Rank &R) { return R.wire; });
    for (Rank &R : d->rk) trp_copy(a_of(R), R.wire, d->nlz, d->npk, d->kmax, 0);
}

void burgers(D *d, int dir, int ivel, int nx, int ny, int nz, double nu, const double *s, const double *u, double *res, double *tmp) {
    ok(tlab_opr_burgers(dir, d->g[dir - 1], ivel, nx, ny, nz, 0, nu, s, u, res, tmp, 0), "tlab_opr_burgers");
}
void partial(D *d, int dir, int nx, int ny, int nz, const double *u, double *res) {
    ok(tlab_opr_partial(dir, d->g[dir - 1], TLAB_OPR_P1, nx, ny, nz, 0, u, res, nullptr), "tlab_opr_partial");
}

// OPR_Burgers_X (dir = 1) / OPR_Burgers_Z (dir = 3) through the I- / K-transposition; self_vel: the operand is the advecting velocity, whose
// transposed copy is kept for the later calls (rhs_global_incompressible_1.f90:98-104: tmp4 / tmp6)
template <class FS>
void burgers_t(D *d, int dir, double nu, FS s_of, int res_idx, bool self_vel) {
    const int csize = dir == 1 ? d->npi : d->npk;
    if (csize == 1) {       // not decomposed in this direction
        for (Rank &R : d->rk) burgers(d, dir, self_vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, d->imax, d->ny, d->kmax, nu, s_of(R), R.q[dir - 1], R.txc[res_idx], R.txc[8]);
        return;
    }
    auto vel = [&](Rank &R) -> double *& { return dir == 1 ? R.u_t : R.w_t; };
    if (dir == 1) trp_i_forward(d, s_of, [&](Rank &R) { return self_vel ? R.u_t : R.ta; });
    else trp_k_forward(d, s_of, [&](Rank &R) { return self_vel ? R.w_t : R.ta; });
    for (Rank &R : d->rk) {
        const double *st = self_vel ? vel(R) : R.ta;
        if (dir == 1) burgers(d, 1, self_vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, d->nx, (int)d->nlx, 1, nu, st, vel(R), R.rt, R.txc[8]);
        else burgers(d, 3, self_vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, (int)d->nlz, 1, d->nzt, nu, st, vel(R), R.rt, R.txc[8]);
    }
    if (dir == 1) trp_i_backward(d, [](Rank &R) { return R.rt; }, [&](Rank &R) { return R.txc[res_idx]; });
    else trp_k_backward(d, [](Rank &R) { return R.rt; }, [&](Rank &R) { return R.txc[res_idx]; });
}
// OPR_Partial_X / _Z (OPR_P1) through the transposition (opr_partial.f90:117-136, :185-195)
void partial_t(D *d, int dir, int src_idx, int dst_idx) {
    const int csize = dir == 1 ? d->npi : d->npk;
    if (csize == 1) {
        for (Rank &R : d->rk) partial(d, dir, d->imax, d->ny, d->kmax, R.txc[src_idx], R.txc[dst_idx]);
        return;
    }
    if (dir == 1) trp_i_forward(d, [&](Rank &R) { return R.txc[src_idx]; }, [](Rank &R) { return R.ta; });
    else trp_k_forward(d, [&](Rank &R) { return R.txc[src_idx]; }, [](Rank &R) { return R.ta; });
    for (Rank &R : d->rk) {
        if (dir == 1) partial(d, 1, d->nx, (int)d->nlx, 1, R.ta, R.rt);
        else partial(d, 3, (int)d->nlz, 1, d->nzt, R.ta, R.rt);
    }
    if (dir == 1) trp_i_backward(d, [](Rank &R) { return R.rt; }, [&](Rank &R) { return R.txc[dst_idx]; });
    else trp_k_backward(d, [](Rank &R) { return R.rt; }, [&](Rank &R) { return R.txc[dst_idx]; });
}

// block txc[idx] (imax, ny, kmax) <-> z-slab (nx, ny, kmax2) of the rank, in place of txc[idx]
void to_slab(D *d, int idx) {
    if (d->npi == 1) return;
    trp_i_forward(d, [&](Rank &R) { return R.txc[idx]; }, [](Rank &R) { return R.tb; });
    for (Rank &R : d->rk) hk(hipMemcpyAsync(R.txc[idx], R.tb, (size_t)d->n * sizeof(double), hipMemcpyDeviceToDevice, tlab_current_stream()), "copy");
}
void to_block(D *d, int idx) {
    if (d->npi == 1) return;
    trp_i_backward(d, [&](Rank &R) { return R.txc[idx]; }, [](Rank &R) { return R.tb; });
    for (Rank &R : d->rk) hk(hipMemcpyAsync(R.txc[idx], R.tb, (size_t)d->n * sizeof(double), hipMemcpyDeviceToDevice, tlab_current_stream()), "copy");
}
int pencil_exchange_start(D *d, bool forward, int pen_idx, int pack_idx);
void pencil_exchange(D *d, bool forward, int pen_idx, int pack_idx) {
    tck(d->tr.wait(d->tr.ctx, (void *)tlab_current_stream(), pencil_exchange_start(d, forward, pen_idx, pack_idx)), "wait");
}
int pencil_exchange_start(D *d, bool forward, int pen_idx, int pack_idx) {
    const int P = d->P;
    std::vector<double *> send, recv;
    std::vector<long long> scnt((size_t)d->rk.size() * P), rcnt((size_t)d->rk.size() * P);
    for (size_t l = 0; l < d->rk.size(); ++l) {
        Rank &R = d->rk[l];
        for (int p = 0; p < P; ++p) {
            const long long mine = 2LL * d->nxl[R.r] * d->ny * d->kmax2, peer = 2LL * d->nxl[p] * d->ny * d->kmax2;
            scnt[l * P + p] = forward ? peer : mine;
            rcnt[l * P + p] = forward ? mine : peer;
        }
        send.push_back(forward ? R.pack[pack_idx] : R.pen[pen_idx]);
        recv.push_back(forward ? R.pen[pen_idx] : R.pack[pack_idx]);
    }
    const int t = d->tr.alltoallv_start(d->tr.ctx, (void *)tlab_current_stream(), 0, send.data(), scnt.data(), recv.data(), rcnt.data());
    tck(t, "alltoallv_start");
    return t;
}
void repack(D *d, Rank &R, double *slab, int pack_idx, int dir) {
    const int P = d->P;
    if (P <= 8) {
        ok(tlab_pencil_repack(slab, R.pack[pack_idx], d->nxh, d->ny, d->kmax2, P, d->ioff.data(), dir), "tlab_pencil_repack");
        return;
    }
    std::vector<long long> base((size_t)P);
    long long off = 0;
    for (int p = 0; p < P; ++p) { base[p] = off; off += (long long)d->nxl[p] * d->ny * d->kmax2; }
    ok(tlab_pencil_repack_blocks(slab, R.pack[pack_idx], d->nxh, d->ny, d->kmax2, P, d->ioff.data(), base.data(), dir), "tlab_pencil_repack_blocks");
}

// OPR_Poisson(.., BCS_NN, ..): forcing in tmp1 (txc[0]), Neumann data in hb / ht; p -> tmp1, dp/dy -> tmp3 (txc[2])
void poisson(D *d) {
    for (Rank &R : d->rk)      // the Neumann data travel in the wall rows of the forcing (opr_elliptic.f90:310-311)
        ok(tlab_pw_set_wall_planes(R.txc[0], R.hb, R.ht, d->imax, d->ny, d->kmax), "tlab_pw_set_wall_planes");
    to_slab(d, 0);
    for (Rank &R : d->rk) {
        ok(tlab_poisson_fft_x(R.poisson, 1, R.txc[0], R.txc[1]), "tlab_poisson_fft_x");
        repack(d, R, R.txc[1], 0, 1);
    }
    pencil_exchange(d, true, 0, 0);
    for (Rank &R : d->rk) {
        ok(tlab_poisson_fft_z(R.poisson, 1, R.pen[0], R.pen[1]), "tlab_poisson_fft_z");
        ok(tlab_poisson_ode(R.poisson, R.pen[1], R.pen[1], R.pen[2]), "tlab_poisson_ode");
        ok(tlab_poisson_fft_z(R.poisson, -1, R.pen[1], R.pen[0]), "tlab_poisson_fft_z");
        ok(tlab_poisson_fft_z(R.poisson, -1, R.pen[2], R.pen[1]), "tlab_poisson_fft_z");
    }
    const int dst_of[2] = {0, 2};
    // both inverse exchanges are started before the first is waited for (two pack buffers): the x transform and the I-transposition of p run under the
    // exchange of dp/dy when the overlapped schedule is on
    int tk[2] = {-1, -1};
    if (d->overlap) for (int i = 0; i < 2; ++i) tk[i] = pencil_exchange_start(d, false, i, i);
    for (int i = 0; i < 2; ++i) {
        const int pk = d->overlap ? i : 0;
        if (d->overlap) tck(d->tr.wait(d->tr.ctx, (void *)tlab_current_stream(), tk[i]), "wait");
        else pencil_exchange(d, false, i, 0);
        for (Rank &R : d->rk) {
            repack(d, R, R.txc[1], pk, -1);
            ok(tlab_poisson_fft_x(R.poisson, -1, R.txc[1], R.txc[dst_of[i]]), "tlab_poisson_fft_x");
        }
        to_block(d, dst_of[i]);
    }
}

void need_bound(D *d) {
    for (Rank &R : d->rk)
        if (!R.bound) throw Fail(TLAB_EINVAL, "tlab_pencil_dns: the arrays of every local rank must be bound first (tlab_pencil_dns_bind)");
}

// the tail of the velocities: sub3 + wall planes (+ Neumann planes) as the reference does it, or -- inside tlab_pencil_dns_substep with Dirichlet walls -- one
// fused pass per component that also does the Runge-Kutta update (rhs_global_incompressible_1.f90:348-375, time.f90:645-664, :272-297)
void finish_velocities(D *d) {
    const int nx = d->imax, ny = d->ny, kmax = d->kmax;
    const long long n = d->n;
    static const bool fused_ok = [] { const char *e = getenv("TLAB_PENCIL_FINAL"); return !(e && atoi(e) == 0); }();
    bool dirichlet = true;
    for (int i = 0; i < 3; ++i) dirichlet = dirichlet && d->flow_jmin[i] != TLAB_DNS_BCS_NEUMANN && d->flow_jmax[i] != TLAB_DNS_BCS_NEUMANN;
    d->fin.done = false;
    if (d->fin.on && fused_ok && dirichlet) {
        for (Rank &R : d->rk)
            for (int i = 0; i < 3; ++i)
                ok(tlab_pw_final_update(R.q[i], R.hq[i], R.txc[1 + i], nullptr, nullptr, d->fin.dte, d->fin.kco, d->fin.scale, nx, ny, kmax), "tlab_pw_final_update");
        d->fin.done = true;
        return;
    }
    for (Rank &R : d->rk) ok(tlab_pw_sub3(R.hq[0], R.hq[1], R.hq[2], R.txc[1], R.txc[2], R.txc[3], n), "tlab_pw_sub3");
    for (Rank &R : d->rk)
        for (int i = 0; i < 3; ++i) {
            const int ibc = (d->flow_jmin[i] == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (d->flow_jmax[i] == TLAB_DNS_BCS_NEUMANN ? 2 : 0);
            if (ibc) ok(tlab_boundary_bcs_neumann_y(d->g[1], ibc, nx, ny, kmax, R.hq[i], R.hb, R.ht, R.txc[0]), "tlab_boundary_bcs_neumann_y");
            ok(tlab_pw_set_wall_planes(R.hq[i], (ibc & 1) ? R.hb : nullptr, (ibc & 2) ? R.ht : nullptr, nx, ny, kmax), "tlab_pw_set_wall_planes");
        }
}

// tools/dns/rhs_global_incompressible_1.f90:98-398
void rhs(D *d, double dte) {
    need_bound(d);
    if (tlab_internal_anelastic() || tlab_internal_dealiasing())
        throw Fail(TLAB_EUNSUPPORTED, "tlab_pencil_dns_rhs: the anelastic formulation / dealiasing filters are not built into the decomposed drivers");
    const int nx = d->imax, ny = d->ny, kmax = d->kmax, ns = d->nscal;
    const long long n = d->n;
    const double nu = d->visc;
    d->fresh = false;      // (this driver follows the reference literally: the tendencies are added to arrays the caller zeroed, time.f90:212-216)
    auto U = [](int i) { return [i](Rank &R) { return R.q[i]; }; };
    auto burgers_y = [&](int ivel, double nu_, auto s_of, int res_idx) {
        for (Rank &R : d->rk) burgers(d, 2, ivel, nx, ny, kmax, nu_, s_of(R), R.q[1], R.txc[res_idx], R.txc[8]);
    };
    auto add3 = [&](auto h_of, int a, int b, int c) {
        for (Rank &R : d->rk) ok(tlab_pw_add3(h_of(R), R.txc[a], R.txc[b], R.txc[c], n), "tlab_pw_add3");
    };
    burgers_t(d, 1, nu, U(0), 0, true);                    // :98   tmp1, u transposed kept
    burgers_y(TLAB_OPR_B_SELF, nu, U(1), 1);               // :99
    burgers_t(d, 3, nu, U(2), 2, true);                    // :100  tmp3, w transposed kept
    burgers_y(TLAB_OPR_B_U_IN, nu, U(0), 6);               // :103
    burgers_t(d, 3, nu, U(0), 7, false);                   // :104
    add3([](Rank &R) { return R.hq[0]; }, 0, 6, 7);
    burgers_t(d, 1, nu, U(1), 6, false);                   // :115
    burgers_t(d, 3, nu, U(1), 7, false);                   // :116
    add3([](Rank &R) { return R.hq[1]; }, 1, 6, 7);
    burgers_t(d, 1, nu, U(2), 6, false);                   // :127
    burgers_y(TLAB_OPR_B_U_IN, nu, U(2), 7);               // :128
    add3([](Rank &R) { return R.hq[2]; }, 2, 6, 7);
    for (int i = 0; i < ns; ++i) {                         // :149-162
        const double kap = d->visc / d->schmidt[i];
        auto sc = [i](Rank &R) { return R.s[i]; };
        burgers_t(d, 1, kap, sc, 0, false);
        burgers_y(TLAB_OPR_B_U_IN, kap, sc, 1);
        burgers_t(d, 3, kap, sc, 2, false);
        add3([i](Rank &R) { return R.hs[i]; }, 0, 1, 2);
    }
    // pressure (:188-260)
    for (Rank &R : d->rk)
        ok(tlab_pw_axpy3(R.txc[1], R.txc[2], R.txc[3], R.hq[1], R.hq[0], R.hq[2], R.q[1], R.q[0], R.q[2], 1.0 / dte, n), "tlab_pw_axpy3");
    for (Rank &R : d->rk) partial(d, 2, nx, ny, kmax, R.txc[1], R.txc[0]);      // :228
    partial_t(d, 1, 2, 1);                                                        // :229
    partial_t(d, 3, 3, 2);                                                        // :230
    for (Rank &R : d->rk) ok(tlab_pw_sum3(R.txc[0], R.txc[1], R.txc[2], n), "tlab_pw_sum3");
    for (Rank &R : d->rk) ok(tlab_pw_get_wall_planes(R.hq[1], R.hb, R.ht, nx, ny, kmax), "tlab_pw_get_wall_planes");
    poisson(d);                                                                   // :284
    partial_t(d, 1, 0, 1);                                                        // :319
    partial_t(d, 3, 0, 3);                                                        // :320
    // boundary conditions (:360-398); y is never split, BOUNDARY_BCS_NEUMANN_Y needs no communication
    finish_velocities(d);
    for (Rank &R : d->rk) {
        auto walls = [&](double *h, int tmin, int tmax) {
            const int ibc = (tmin == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (tmax == TLAB_DNS_BCS_NEUMANN ? 2 : 0);
            if (ibc) ok(tlab_boundary_bcs_neumann_y(d->g[1], ibc, nx, ny, kmax, h, R.hb, R.ht, R.txc[0]), "tlab_boundary_bcs_neumann_y");
            ok(tlab_pw_set_wall_planes(h, (ibc & 1) ? R.hb : nullptr, (ibc & 2) ? R.ht : nullptr, nx, ny, kmax), "tlab_pw_set_wall_planes");
        };
        for (int i = 0; i < ns; ++i) walls(R.hs[i], d->scal_jmin[i], d->scal_jmax[i]);
    }
}

// ---- the same RHS with the transpositions started AHEAD of independent launches (missing 2; the reference's analogue:
// tools/dns/rhs_global_incompressible_nbc.f90:135-382).  The exchanges of the transport run on its own stream between start and wait, so what is
// issued on the compute stream in between overlaps them: while operator i is applied to the transposed field, the forward transposition of field
// i + 1 and the backward transposition of result i - 1 are in flight, and the local y operators, the sums and the packing passes fill the rest.
// Same kernels with the same arguments as rhs(): results are bit-identical (tests/test_gpu_pencil.py).
struct TOp {                         // one operator along x (dir 1) or z (dir 3) through its transposition
    int dir;
    bool burg, self_vel;
    double nu;
    std::function<double *(Rank &)> src;
    std::function<double *(Rank &)> dst;
    int t_f = -1, t_b = -1;
    long long mark_f = 0, mark_b = 0;  // d->launches when the exchange was started
    int needs_local = -1;            // index of the launch in the local list that must have run before this operator's result is written (the sum that frees its slot)
};
struct LocalOp {                     // launches that touch no transposition; ready once the backward transpositions of ops <= after are complete
    int after;                       // ... and once the local launch `needs` has run (-1: none)
    std::function<void()> run;
    int needs = -1;
    bool done = false;
};

void note(D *d, const char *what, int i = -1) {
    if (what[0] == 'l') ++d->launches;           // "launch ..."
    if (!d->tracing) return;
    d->trace += what;
    if (i >= 0) d->trace += " " + std::to_string(i);
    d->trace += "\n";
}
template <class FS, class FR>
int a2a_start(D *d, int which, long long blk, FS send_of, FR recv_of) {
    const int S = which == 1 ? d->npi : d->npk;
    std::vector<double *> send, recv;
    std::vector<long long> cnt((size_t)d->rk.size() * S, blk);
    for (Rank &R : d->rk) { send.push_back(send_of(R)); recv.push_back(recv_of(R)); }
    const int t = d->tr.alltoallv_start(d->tr.ctx, (void *)tlab_current_stream(), which, send.data(), cnt.data(), recv.data(), cnt.data());
    tck(t, "alltoallv_start");
    return t;
}
void a2a_wait(D *d, int t) { tck(d->tr.wait(d->tr.ctx, (void *)tlab_current_stream(), t), "wait"); }

void run_pipeline(D *d, std::vector<TOp> &ops, std::vector<LocalOp> &local) {
    const int m = (int)ops.size();
    auto ta_of = [](Rank &R, int b) { return b ? R.tb : R.ta; };
    auto rt_of = [](Rank &R, int b) { return b ? R.rt2 : R.rt; };
    auto wf_of = [](Rank &R, int b) { return b ? R.wire2 : R.wire; };
    auto decomposed = [&](const TOp &o) { return (o.dir == 1 ? d->npi : d->npk) > 1; };
    auto tdst = [&](const TOp &o, Rank &R, int b) -> double * { return o.self_vel ? (o.dir == 1 ? R.u_t : R.w_t) : ta_of(R, b); };
    std::function<bool()> fill_gap = [] { return false; };      // set below: one ready local launch, whatever the reserve policy says
    auto f_start = [&](int i) {
        TOp &o = ops[i];
        if (!decomposed(o)) return;
        const int b = i & 1;
        if (o.dir == 1) {
            o.t_f = a2a_start(d, 1, d->nlx * d->imax, o.src, [&](Rank &R) { return wf_of(R, b); });        // a is blocked by peer as it stands
        } else {
            note(d, "launch pack", i);
            for (Rank &R : d->rk) trp_copy(o.src(R), wf_of(R, b), d->nlz, d->npk, d->kmax, 1);
            o.t_f = a2a_start(d, 2, d->nlz * d->kmax, [&](Rank &R) { return wf_of(R, b); }, [&](Rank &R) { return tdst(o, R, b); });
        }
        note(d, "start forward", i);
        o.mark_f = d->launches;
    };
    auto f_finish = [&](int i) {
        TOp &o = ops[i];
        if (!decomposed(o)) return;
        const int b = i & 1;
        if (d->launches == o.mark_f && !fill_gap()) note(d, "nothing left to overlap");
        a2a_wait(d, o.t_f);
        note(d, "wait forward", i);
        if (o.dir == 1) for (Rank &R : d->rk) trp_copy(tdst(o, R, b), wf_of(R, b), d->imax, d->npi, d->nlx, 0);
    };
    auto apply = [&](int i) {
        TOp &o = ops[i];
        const int b = i & 1;
        note(d, "launch operator", i);
        if (o.needs_local >= 0 && !local[(size_t)o.needs_local].done) throw Fail(TLAB_EINVAL, "internal: pencil schedule would overwrite a term that has not been summed yet");
        for (Rank &R : d->rk) {
            if (!decomposed(o)) {      // not split in this direction: the operator acts on the block itself
                if (o.burg) burgers(d, o.dir, o.self_vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, d->imax, d->ny, d->kmax, o.nu, o.src(R), R.q[o.dir - 1], o.dst(R), R.txc[8]);
                else partial(d, o.dir, d->imax, d->ny, d->kmax, o.src(R), o.dst(R));
                continue;
            }
            const double *st = tdst(o, R, b), *vel = o.dir == 1 ? R.u_t : R.w_t;
            if (o.burg) {
                if (o.dir == 1) burgers(d, 1, o.self_vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, d->nx, (int)d->nlx, 1, o.nu, st, vel, rt_of(R, b), R.txc[8]);
                else burgers(d, 3, o.self_vel ? TLAB_OPR_B_SELF : TLAB_OPR_B_U_IN, (int)d->nlz, 1, d->nzt, o.nu, st, vel, rt_of(R, b), R.txc[8]);
            } else {
                if (o.dir == 1) partial(d, 1, d->nx, (int)d->nlx, 1, st, rt_of(R, b));
                else partial(d, 3, (int)d->nlz, 1, d->nzt, st, rt_of(R, b));
            }
        }
    };
    auto b_start = [&](int i) {
        TOp &o = ops[i];
        if (!decomposed(o)) return;
        const int b = i & 1;
        if (o.dir == 1) {
            note(d, "launch pack", i);
            for (Rank &R : d->rk) trp_copy(rt_of(R, b), R.wireb[b], d->imax, d->npi, d->nlx, 1);
            o.t_b = a2a_start(d, 1, d->nlx * d->imax, [&](Rank &R) { return R.wireb[b]; }, o.dst);
        } else {
            o.t_b = a2a_start(d, 2, d->nlz * d->kmax, [&](Rank &R) { return rt_of(R, b); }, [&](Rank &R) { return R.wireb[b]; });
        }
        note(d, "start backward", i);
        o.mark_b = d->launches;
    };
    auto b_finish = [&](int i) {
        TOp &o = ops[i];
        if (!decomposed(o)) return;
        const int b = i & 1;
        if (d->launches == o.mark_b && !fill_gap()) note(d, "nothing left to overlap");
        a2a_wait(d, o.t_b);
        note(d, "wait backward", i);
        if (o.dir == 3) for (Rank &R : d->rk) trp_copy(o.dst(R), R.wireb[b], d->nlz, d->npk, d->kmax, 0);
    };
    int finished = -1;               // backward transpositions complete up to this operator
    auto ready = [&](const LocalOp &l) { return !l.done && l.after <= finished && (l.needs < 0 || local[(size_t)l.needs].done); };
    auto run_one = [&](LocalOp &l) { note(d, "launch local"); l.run(); l.done = true; };
    // one launch for a gap between a start and its wait.  Only two gaps have no operator launch of their own -- the first forward transposition and
    // the last backward one -- so one launch that depends on nothing is held back for the end (keep_reserve)
    auto run_local = [&](bool keep_reserve) {
        int independent = 0;
        for (const LocalOp &l : local) if (!l.done && l.after < 0 && l.needs < 0) ++independent;
        for (LocalOp &l : local) {
            if (!ready(l)) continue;
            if (keep_reserve && l.after < 0 && l.needs < 0 && independent == 1) continue;
            run_one(l);
            return true;
        }
        return false;
    };
    fill_gap = [&] { return run_local(false); };
    std::function<void(int)> force = [&](int j) {      // a launch an upcoming operator needs (the sum that frees its result slot), with what it needs itself
        if (j < 0 || local[(size_t)j].done) return;
        force(local[(size_t)j].needs);
        if (local[(size_t)j].after > finished) throw Fail(TLAB_EINVAL, "internal: pencil schedule needs a sum whose terms have not arrived");
        run_one(local[(size_t)j]);
    };
    std::vector<char> applied((size_t)m, 0);
    if (m > 0) f_start(0);
    for (int i = 0; i < m; ++i) {
        if (i + 1 < m) {
            f_start(i + 1);                        // the next field is on its way while this one is worked on
            if (!decomposed(ops[i + 1])) {         // ... or, where its direction is not split, its operator is an independent launch: now
                force(ops[i + 1].needs_local);
                apply(i + 1);
                applied[(size_t)i + 1] = 1;
            }
        }
        if (i == 0) run_local(true);               // (the first forward transposition has no operator launch before its wait; f_finish fills the gap in any case)
        f_finish(i);
        if (!applied[(size_t)i]) { force(ops[i].needs_local); apply(i); }
        b_start(i);
        if (i > 0) { run_local(true); b_finish(i - 1); finished = i - 1; }
    }
    if (m > 0) {
        b_finish(m - 1);
        finished = m - 1;
    }
    for (bool any = true; any;) {                   // the rest, in dependency order
        any = false;
        for (LocalOp &l : local) if (ready(l)) { run_one(l); any = true; }
    }
    for (LocalOp &l : local) if (!l.done) throw Fail(TLAB_EINVAL, "internal: pencil schedule left a launch behind");
}

void rhs_overlapped(D *d, double dte) {
    need_bound(d);
    if (tlab_internal_anelastic() || tlab_internal_dealiasing())
        throw Fail(TLAB_EUNSUPPORTED, "tlab_pencil_dns_rhs: the anelastic formulation / dealiasing filters are not built into the decomposed drivers");
    const int nx = d->imax, ny = d->ny, kmax = d->kmax, ns = d->nscal;
    const long long n = d->n;
    const double nu = d->visc;
    d->fresh = false;
    d->trace.clear();
    // result slots: three per equation, from three sets in rotation (the sums free a set two equations before it is written again)
    auto slot = [](int eq, int k) { return [eq, k](Rank &R) -> double * { const int s = eq % 3; return s == 0 ? R.txc[k] : s == 1 ? R.txc[3 + k] : (k < 2 ? R.txc[6 + k] : R.tx); }; };
    auto U = [](int i) { return [i](Rank &R) { return R.q[i]; }; };
    std::vector<TOp> ops;
    std::vector<LocalOp> local;
    auto top = [&](int dir, bool self_vel, double nu_, std::function<double *(Rank &)> src, std::function<double *(Rank &)> dst) {
        TOp o; o.dir = dir; o.burg = true; o.self_vel = self_vel; o.nu = nu_; o.src = src; o.dst = dst;
        ops.push_back(o);
        return (int)ops.size() - 1;
    };
    std::vector<int> sum_after, sum_index;      // per equation: the operator whose backward transposition completes its three terms; its sum in `local`
    std::vector<int> y_index;
    auto ylocal = [&](int ivel, double nu_, std::function<double *(Rank &)> s_of, std::function<double *(Rank &)> dst, int eq) {
        // the y term of equation eq writes into the set equation eq - 3 summed from: not before that sum
        if ((int)y_index.size() <= eq) y_index.resize(eq + 1, -1);
        y_index[eq] = (int)local.size();
        LocalOp l{eq >= 3 ? sum_after[eq - 3] : -1, [=]() { for (Rank &R : d->rk) burgers(d, 2, ivel, nx, ny, kmax, nu_, s_of(R), R.q[1], dst(R), R.txc[8]); }};
        l.needs = eq >= 3 ? sum_index[eq - 3] : -1;
        local.push_back(l);
    };
    // (k0, k1, k2: the order of the three terms in the reference's sum, rhs_global_incompressible_1.f90:106-112, :118-124, :130-136: the self-advection term first)
    auto sum = [&](int after, std::function<double *(Rank &)> h_of, int eq, int k0 = 0, int k1 = 1, int k2 = 2) {
        if ((int)sum_after.size() <= eq) { sum_after.resize(eq + 1, -1); sum_index.resize(eq + 1, -1); }
        sum_after[eq] = after;
        sum_index[eq] = (int)local.size();
        LocalOp l{after, [=]() { for (Rank &R : d->rk) ok(tlab_pw_add3(h_of(R), slot(eq, k0)(R), slot(eq, k1)(R), slot(eq, k2)(R), n), "tlab_pw_add3"); }};
        l.needs = (int)y_index.size() > eq ? y_index[eq] : -1;      // its own y term
        local.push_back(l);
    };
    // equations 0, 1, 2 = u, v, w; slot 0 / 1 / 2 of an equation = its x / y / z Burgers term.  The two self-advecting transposed operators come first:
    // they leave the transposed u and w every other operator along x / z needs (rhs_global_incompressible_1.f90:98-104)
    const int o_xu = top(1, true, nu, U(0), slot(0, 0));                                   // :98
    const int o_zw = top(3, true, nu, U(2), slot(2, 2));                                   // :100
    ylocal(TLAB_OPR_B_SELF, nu, U(1), slot(1, 1), 1);                                      // :99
    ylocal(TLAB_OPR_B_U_IN, nu, U(0), slot(0, 1), 0);                                      // :103
    const int o_zu = top(3, false, nu, U(0), slot(0, 2));                                  // :104
    sum(std::max(o_xu, o_zu), [](Rank &R) { return R.hq[0]; }, 0);
    const int o_xv = top(1, false, nu, U(1), slot(1, 0));                                  // :115
    const int o_zv = top(3, false, nu, U(1), slot(1, 2));                                  // :116
    sum(std::max(o_xv, o_zv), [](Rank &R) { return R.hq[1]; }, 1, 1, 0, 2);                // tmp2 (y, self) + tmp7 (x) + tmp8 (z)
    const int o_xw = top(1, false, nu, U(2), slot(2, 0));                                  // :127
    ylocal(TLAB_OPR_B_U_IN, nu, U(2), slot(2, 1), 2);                                      // :128
    sum(std::max(o_zw, o_xw), [](Rank &R) { return R.hq[2]; }, 2, 2, 0, 1);                // tmp3 (z, self) + tmp7 (x) + tmp8 (y)
    for (int i = 0; i < ns; ++i) {                                                         // :149-162
        const double kap = d->visc / d->schmidt[i];
        auto sc = [i](Rank &R) { return R.s[i]; };
        const int a = top(1, false, kap, sc, slot(3 + i, 0));
        ops[a].needs_local = sum_index[i];
        ylocal(TLAB_OPR_B_U_IN, kap, sc, slot(3 + i, 1), 3 + i);
        const int b = top(3, false, kap, sc, slot(3 + i, 2));
        ops[b].needs_local = sum_index[i];
        sum(std::max(a, b), [i](Rank &R) { return R.hs[i]; }, 3 + i);
    }
    // (the y terms of equation k + 3 write the set of equation k: they must not run before its sum; the list order of `local` guarantees it, every
    // sum of an earlier equation standing before the y term of a later one that reuses its set)
    run_pipeline(d, ops, local);
    // pressure (:188-260)
    for (Rank &R : d->rk)
        ok(tlab_pw_axpy3(R.txc[1], R.txc[2], R.txc[3], R.hq[1], R.hq[0], R.hq[2], R.q[1], R.q[0], R.q[2], 1.0 / dte, n), "tlab_pw_axpy3");
    {
        std::vector<TOp> po;
        std::vector<LocalOp> pl;
        TOp a; a.dir = 1; a.burg = false; a.self_vel = false; a.nu = 0.0; a.src = [](Rank &R) { return R.txc[2]; }; a.dst = [](Rank &R) { return R.txc[4]; };      // :229
        TOp b; b.dir = 3; b.burg = false; b.self_vel = false; b.nu = 0.0; b.src = [](Rank &R) { return R.txc[3]; }; b.dst = [](Rank &R) { return R.txc[5]; };      // :230
        po.push_back(a); po.push_back(b);
        pl.push_back({-1, [=]() { for (Rank &R : d->rk) partial(d, 2, nx, ny, kmax, R.txc[1], R.txc[0]); }});                                                  // :228
        pl.push_back({-1, [=]() { for (Rank &R : d->rk) ok(tlab_pw_get_wall_planes(R.hq[1], R.hb, R.ht, nx, ny, kmax), "tlab_pw_get_wall_planes"); }});
        run_pipeline(d, po, pl);
    }
    for (Rank &R : d->rk) ok(tlab_pw_sum3(R.txc[0], R.txc[4], R.txc[5], n), "tlab_pw_sum3");
    poisson(d);                                                                   // :284
    {
        std::vector<TOp> po;
        std::vector<LocalOp> pl;
        TOp a; a.dir = 1; a.burg = false; a.self_vel = false; a.nu = 0.0; a.src = [](Rank &R) { return R.txc[0]; }; a.dst = [](Rank &R) { return R.txc[1]; };      // :319
        TOp b; b.dir = 3; b.burg = false; b.self_vel = false; b.nu = 0.0; b.src = [](Rank &R) { return R.txc[0]; }; b.dst = [](Rank &R) { return R.txc[3]; };      // :320
        po.push_back(a); po.push_back(b);
        // the wall conditions of the scalars (:379-396) do not involve the pressure: they fill the gaps of these two transpositions (scratch: txc[4], hb / ht
        // are free again after the solver)
        for (int i = 0; i < ns; ++i)
            pl.push_back({-1, [=]() {
                for (Rank &R : d->rk) {
                    const int ibc = (d->scal_jmin[i] == TLAB_DNS_BCS_NEUMANN ? 1 : 0) + (d->scal_jmax[i] == TLAB_DNS_BCS_NEUMANN ? 2 : 0);
                    if (ibc) ok(tlab_boundary_bcs_neumann_y(d->g[1], ibc, nx, ny, kmax, R.hs[i], R.hb, R.ht, R.txc[4]), "tlab_boundary_bcs_neumann_y");
                    ok(tlab_pw_set_wall_planes(R.hs[i], (ibc & 1) ? R.hb : nullptr, (ibc & 2) ? R.ht : nullptr, nx, ny, kmax), "tlab_pw_set_wall_planes");
                }
            }});
        run_pipeline(d, po, pl);
    }
    finish_velocities(d);
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

int tlab_pencil_transport_loopback(tlab_pencil_transport *out, int npro_i, int npro_k) {
    if (!out || npro_i < 1 || npro_k < 1) { tlab_set_error("tlab_pencil_transport_loopback: bad arguments"); return TLAB_EINVAL; }
    out->ctx = new Loopback{npro_i, npro_k};
    out->npro_i = npro_i; out->npro_k = npro_k; out->nlocal = npro_i * npro_k; out->first = 0;
    out->alltoallv_start = lb_a2a; out->wait = lb_wait; out->allreduce = lb_allreduce; out->destroy = lb_destroy;
    return TLAB_OK;
}

int tlab_pencil_dns_create(tlab_pencil_dns_t *out, const tlab_pencil_transport *tr, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx,
                           int ny, int nz_total, int nscal, double visc, const double *schmidt) {
    return guarded([&] {
        if (!out || !tr || !gx || !gy || !gz || nscal < 0 || (nscal > 0 && !schmidt) || visc <= 0.0) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_create: bad arguments");
        if (!tr->alltoallv_start || !tr->wait || !tr->allreduce) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_create: incomplete transport");
        const int npi = tr->npro_i, npk = tr->npro_k, P = npi * npk;
        if (npi < 1 || npk < 1 || tr->nlocal < 1 || tr->first < 0 || tr->first + tr->nlocal > P) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_create: rank layout");
        // the conditions of the reference's decomposition, checked before anything is allocated
        if (nx % npi || nz_total % npk) throw Fail(TLAB_EINVAL, "nx, nz must be divisible by npro_i, npro_k");
        const int imax = nx / npi, kmax = nz_total / npk;
        if (imax % 2) throw Fail(TLAB_EINVAL, "imax must be even (opr_fourier.f90:73-76)");
        if (kmax % npi) throw Fail(TLAB_EINVAL, "npro_i must divide kmax: the x lines of a rank after the I-transposition form whole z planes");
        if (((long long)imax * ny) % npk) throw Fail(TLAB_EINVAL, "imax*jmax must be divisible by npro_k (tlab_mpi_transpose.f90:292)");
        if ((nx / 2 + 1) / P < 1) throw Fail(TLAB_EINVAL, "fewer kx modes than ranks");
        if (P > 16) throw Fail(TLAB_EUNSUPPORTED, "tlab_pencil_dns_create: at most 16 ranks (one node)");
        if (tlab_internal_anelastic() || tlab_internal_dealiasing())
            throw Fail(TLAB_EUNSUPPORTED, "tlab_pencil_dns_create: the anelastic formulation / dealiasing filters are not built into the decomposed drivers");
        auto d = std::make_unique<tlab_pencil_dns>();
        d->g[0] = gx; d->g[1] = gy; d->g[2] = gz;
        if (const char *e = getenv("TLAB_PENCIL_OVERLAP")) d->overlap = atoi(e) != 0;
        d->npi = npi; d->npk = npk; d->P = P; d->nx = nx; d->ny = ny; d->nzt = nz_total; d->imax = imax; d->kmax = kmax; d->kmax2 = kmax / npi;
        d->nxh = nx / 2 + 1; d->nscal = nscal; d->visc = visc;
        d->n = (long long)imax * ny * kmax;
        d->npage_i = (long long)ny * kmax; d->nlx = (long long)ny * d->kmax2;
        d->npage_k = (long long)imax * ny; d->nlz = d->npage_k / npk;
        d->isize_txc = (long long)(nx + 2) * ny * d->kmax2;
        if (nscal) d->schmidt.assign(schmidt, schmidt + nscal);
        d->scal_jmin.assign(nscal, TLAB_DNS_BCS_DIRICHLET);
        d->scal_jmax.assign(nscal, TLAB_DNS_BCS_DIRICHLET);
        const int base = d->nxh / P, rem = d->nxh % P;
        for (int r = 0; r < P; ++r) { d->nxl.push_back(base + (r < rem ? 1 : 0)); d->ioff.push_back(r * base + std::min(r, rem)); }
        d->rk.resize(tr->nlocal);
        for (int l = 0; l < tr->nlocal; ++l) {
            Rank &R = d->rk[l];
            R.r = tr->first + l; R.pi = R.r % npi; R.pk = R.r / npi;
            ok(tlab_poisson_plan_create_pencil(&R.poisson, gx, gy, gz, nx, ny, d->kmax2, nz_total, d->ioff[R.r], d->nxl[R.r]), "tlab_poisson_plan_create_pencil");
            R.hb = dalloc((size_t)imax * kmax); R.ht = dalloc((size_t)imax * kmax);
            for (double **p : {&R.rt, &R.u_t, &R.w_t, &R.ta, &R.tb, &R.wire}) *p = dalloc((size_t)d->n);
            if (d->overlap) for (double **p : {&R.rt2, &R.wire2, &R.wireb[0], &R.wireb[1], &R.tx}) *p = dalloc((size_t)d->n);
            for (int i = 0; i < 3; ++i) R.pen[i] = dalloc((size_t)2 * d->nxl[R.r] * ny * nz_total);
            for (int i = 0; i < 2; ++i) R.pack[i] = dalloc((size_t)2 * d->nxh * ny * d->kmax2);
        }
        d->tr = *tr;      // ownership of the transport context passes here, on success only
        *out = d.release();
    });
}

int tlab_pencil_dns_destroy(tlab_pencil_dns_t d) {
    (void)tlab_internal_deferred_flush();
    if (d) (void)hipDeviceSynchronize();
    delete d;
    return TLAB_OK;
}

int tlab_pencil_dns_bind(tlab_pencil_dns_t d, int l, double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc) {
    return guarded([&] {
        if (!d || l < 0 || l >= (int)d->rk.size() || !q || !hq || !txc || (d->nscal > 0 && (!s || !hs))) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_bind: bad arguments");
        Rank &R = d->rk[l];
        R.q.assign(q, q + 3); R.hq.assign(hq, hq + 3); R.txc.assign(txc, txc + 9);
        R.s.assign(s, s + d->nscal); R.hs.assign(hs, hs + d->nscal);
        for (double *p : R.q) if (!p) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_bind: null array");
        for (double *p : R.hq) if (!p) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_bind: null array");
        for (double *p : R.txc) if (!p) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_bind: null array");
        R.bound = true;
    });
}

long long tlab_pencil_dns_info(tlab_pencil_dns_t d, int what) {
    if (!d) return TLAB_EINVAL;
    switch (what) {
    case 0: return d->imax;
    case 1: return d->kmax;
    case 2: return d->kmax2;
    case 3: return d->isize_txc;
    case 4: return (long long)d->rk.size();
    case 5: return d->rk.empty() ? 0 : d->rk[0].r;
    }
    return TLAB_EINVAL;
}

int tlab_pencil_dns_set_bcs(tlab_pencil_dns_t d, const int *flow_jmin, const int *flow_jmax, const int *scal_jmin, const int *scal_jmax) {
    return guarded([&] {
        if (!d || !flow_jmin || !flow_jmax || (d->nscal > 0 && (!scal_jmin || !scal_jmax))) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_set_bcs: bad arguments");
        auto valid = [](int t) { return t == TLAB_DNS_BCS_DIRICHLET || t == TLAB_DNS_BCS_NEUMANN; };
        for (int i = 0; i < 3; ++i)
            if (!valid(flow_jmin[i]) || !valid(flow_jmax[i])) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_set_bcs: type must be DNS_BCS_DIRICHLET or DNS_BCS_NEUMANN");
        for (int i = 0; i < d->nscal; ++i)
            if (!valid(scal_jmin[i]) || !valid(scal_jmax[i])) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_set_bcs: type must be DNS_BCS_DIRICHLET or DNS_BCS_NEUMANN");
        if (flow_jmin[1] != TLAB_DNS_BCS_DIRICHLET || flow_jmax[1] != TLAB_DNS_BCS_DIRICHLET)
            throw Fail(TLAB_EUNSUPPORTED, "tlab_pencil_dns_set_bcs: the wall-normal velocity must be Dirichlet (impermeable walls; the pressure BCs assume v = 0)");
        for (int i = 0; i < 3; ++i) { d->flow_jmin[i] = flow_jmin[i]; d->flow_jmax[i] = flow_jmax[i]; }
        for (int i = 0; i < d->nscal; ++i) { d->scal_jmin[i] = scal_jmin[i]; d->scal_jmax[i] = scal_jmax[i]; }
    });
}

// hq = hs = 0 of TIME_RUNGEKUTTA (time.f90:212-216): this driver adds to the tendencies like the reference, so the start of a step zeroes them
int tlab_pencil_dns_begin_step(tlab_pencil_dns_t d) {
    return guarded([&] {
        if (!d) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_begin_step: null handle");
        need_bound(d);
        for (Rank &R : d->rk) {
            for (double *h : R.hq) ok(tlab_pw_fill(h, 0.0, d->n), "tlab_pw_fill");
            for (double *h : R.hs) ok(tlab_pw_fill(h, 0.0, d->n), "tlab_pw_fill");
        }
    });
}

// the order in which the last RHS started exchanges, issued launches and waited (one event per line): tests assert that work stands between every
// start and its wait.  on != 0 switches the recording on (and returns the text recorded so far), 0 off.
int tlab_pencil_dns_trace(tlab_pencil_dns_t d, int on, char *buf, int size) {
    if (!d) return TLAB_EINVAL;
    d->tracing = on != 0;
    if (buf && size > 0) {
        const size_t k = std::min((size_t)size - 1, d->trace.size());
        std::memcpy(buf, d->trace.data(), k);
        buf[k] = 0;
    }
    return TLAB_OK;
}

int tlab_pencil_dns_rhs(tlab_pencil_dns_t d, double dte) {
    return guarded([&] {
        if (!d || !(dte > 0.0)) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_rhs: bad arguments");
        if (d->overlap) rhs_overlapped(d, dte); else rhs(d, dte);
    });
}

int tlab_pencil_dns_substep(tlab_pencil_dns_t d, double dte, double kco, int scale_tendencies) {
    return guarded([&] {
        if (!d || !(dte > 0.0)) throw Fail(TLAB_EINVAL, "tlab_pencil_dns_substep: bad arguments");
        d->fin.on = true; d->fin.dte = dte; d->fin.kco = kco; d->fin.scale = scale_tendencies;
        struct Off { D *d; ~Off() { d->fin.on = false; } } off{d};      // (also when the RHS throws)
        if (d->overlap) rhs_overlapped(d, dte); else rhs(d, dte);
        for (Rank &R : d->rk) {      // time.f90:645-664, :272-297
            for (int i = 0; i < 3 && !d->fin.done; ++i) ok(tlab_pw_rk_update(R.q[i], R.hq[i], dte, kco, scale_tendencies, d->n), "tlab_pw_rk_update");
            for (int i = 0; i < d->nscal; ++i) ok(tlab_pw_rk_update(R.s[i], R.hs[i], dte, kco, scale_tendencies, d->n), "tlab_pw_rk_update");
        }
    });
}

}  // extern "C"

// deferred.cpp: the arrays of the ONE local rank of a Fortran / MPI host
bool tlab_internal_pencil_bound(tlab_pencil_dns_t d, double *const **q, double *const **s, double *const **hq, double *const **hs, int *nscal, long long *n) {
    if (!d || d->rk.size() != 1 || !d->rk[0].bound) return false;
    *q = d->rk[0].q.data(); *s = d->rk[0].s.data(); *hq = d->rk[0].hq.data(); *hs = d->rk[0].hs.data();
    *nscal = d->nscal; *n = d->n;
    return true;
}
