// The tail of the Runge-Kutta substep for a host whose time loop is NOT patched (include/tlab_amd.h: tlab_deferred_*).
//
// tools/dns/time.f90 of the reference runs, per substep,
//     call RHS_GLOBAL_INCOMPRESSIBLE_1()                          (:612, link-time replacement: the whole assembly on the device)
//     call DAXPY(n, dte, hq(1,is), 1, q(1,is), 1)   is = 1..3     (:649-660, the -DUSE_BLAS branches)   q += dte hq
//     call DAXPY(n, dte, hs(1,is), 1, s(1,is), 1)   is = 1..ns
//     call DSCAL(n, kco, hq(1,is), 1) ...                         (:279-293, every substep but the last)  hq *= kco
// and `hq = 0 ; hs = 0` at the start of a step (:212-216).  Executed one by one these are 2 (3 + ns) extra passes over the fields per substep
// (4.5 ms of 20.8 at 512^3), because the last kernels of the device RHS that hold each finished tendency in registers cannot know dte's partner
// kco yet.  They CAN when the calls are only recorded: this layer keeps the RHS call and the BLAS calls that follow it as a description, and when
// the description is complete -- or anything else wants the stream -- runs the ONE fused call tlab_time_substep_incompressible_explicit(dte, kco,
// scale) that the patched host of INTEGRATION.md section 3b would have made.  Same kernels, same arguments: the fields are those of the fused
// route to the bit.  A sequence that does not match (other vectors, other factors, another order) is executed literally, in the order it came.
//
// What makes it safe: every launch of the library fetches its stream through tlab_current_stream(), which flushes first; tlab_sync, the copies and
// tlab_free flush as well.  What it cannot see: a host statement that reads a device array directly (hipMalloc memory is host-addressable on
// MI355X): such a host calls tlab_deferred_flush() or tlab_sync() first, or keeps the layer off (the default).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/tlab_amd.h"

extern void tlab_set_error(const std::string &s);
long long tlab_internal_dns_points(tlab_dns_t d);      // rhs.cpp
int tlab_internal_dns_nscal(tlab_dns_t d);
// the arrays a decomposed driver is bound to (one local rank: a Fortran / MPI host), slab.cpp / pencil.cpp; false: not bound, or several local ranks
bool tlab_internal_slab_bound(tlab_slab_dns_t d, double *const **q, double *const **s, double *const **hq, double *const **hs, int *nscal, long long *n);
bool tlab_internal_pencil_bound(tlab_pencil_dns_t d, double *const **q, double *const **s, double *const **hq, double *const **hs, int *nscal, long long *n);

namespace {
struct Range { double *p; long long n; };
struct Pending {
    bool rhs = false;
    int kind = 0;                                 // 0: tlab_dns (one domain), 1: tlab_slab_dns, 2: tlab_pencil_dns -- the same tail behind each of them
    tlab_dns_t d = nullptr;
    tlab_slab_dns_t slab = nullptr;
    tlab_pencil_dns_t pencil = nullptr;
    double dte = 0.0, kco = 1.0;
    int nf = 0;                                   // 3 + nscal
    long long n = 0;
    std::vector<double *> q, s, hq, hs, txc;      // as handed to the RHS
    std::vector<double *> x, y;                   // per field: tendency, state
    std::vector<char> upd, scl;
    int nupd = 0, nscl = 0;
    std::vector<Range> zeros;                     // `hq = 0` of the start of a step, not yet executed
};
Pending g_p;
bool g_on = false, g_busy = false;
long long g_stat[6] = {0, 0, 0, 0, 0, 0};        // fused substeps, literal flushes, begin_steps, eager axpy, eager scal, eager zero

struct Busy {
    bool was;
    Busy() : was(g_busy) { g_busy = true; }
    ~Busy() { g_busy = was; }
};

int run_zeros_eagerly() {
    std::vector<Range> z;
    z.swap(g_p.zeros);
    for (const Range &r : z) {
        ++g_stat[5];
        const int rc = tlab_pw_fill(r.p, 0.0, r.n);
        if (rc != TLAB_OK) return rc;
    }
    return TLAB_OK;
}

// do the pending zero fills cover exactly the tendencies of the pending RHS?  (then they are the `hq = 0 ; hs = 0` of time.f90:212-216 and become
// tlab_dns_begin_step: the first launch of each field overwrites instead of accumulating)
bool zeros_are_the_tendencies(const Pending &p) {
    if (p.zeros.empty()) return false;
    long long total = 0;
    for (const Range &r : p.zeros) total += r.n;
    if (total != (long long)p.nf * p.n) return false;
    for (int f = 0; f < p.nf; ++f) {
        bool in = false;
        for (const Range &r : p.zeros) in = in || (p.x[f] >= r.p && p.x[f] + p.n <= r.p + r.n);
        if (!in) return false;
    }
    return true;
}

int run_begin(const Pending &p) {
    return p.kind == 0 ? tlab_dns_begin_step(p.d) : p.kind == 1 ? tlab_slab_dns_begin_step(p.slab) : tlab_pencil_dns_begin_step(p.pencil);
}
int run_substep(Pending &p, double kco, int scale) {
    if (p.kind == 1) return tlab_slab_dns_substep(p.slab, p.dte, kco, scale);
    if (p.kind == 2) return tlab_pencil_dns_substep(p.pencil, p.dte, kco, scale);
    return tlab_time_substep_incompressible_explicit(p.d, p.dte, kco, scale, p.q.data(), p.s.empty() ? nullptr : p.s.data(), p.hq.data(),
                                                     p.hs.empty() ? nullptr : p.hs.data(), p.txc.data());
}
int run_rhs(Pending &p) {
    if (p.kind == 1) return tlab_slab_dns_rhs(p.slab, p.dte);
    if (p.kind == 2) return tlab_pencil_dns_rhs(p.pencil, p.dte);
    return tlab_rhs_global_incompressible_1(p.d, p.dte, p.q.data(), p.s.empty() ? nullptr : p.s.data(), p.hq.data(), p.hs.empty() ? nullptr : p.hs.data(),
                                            p.txc.data());
}

int flush_impl() {
    if (g_busy) return TLAB_OK;
    Busy b;
    if (!g_p.rhs) return run_zeros_eagerly();
    Pending p;
    std::swap(p, g_p);                 // whatever runs below sees an empty description
    int rc = TLAB_OK;
    if (zeros_are_the_tendencies(p)) {
        ++g_stat[2];
        rc = run_begin(p);
    } else {
        for (const Range &r : p.zeros) {
            ++g_stat[5];
            if (rc == TLAB_OK) rc = tlab_pw_fill(r.p, 0.0, r.n);
        }
    }
    if (rc != TLAB_OK) return rc;
    if (p.nupd == p.nf && (p.nscl == 0 || p.nscl == p.nf)) {      // the whole substep, as the patched host would have called it
        ++g_stat[0];
        return run_substep(p, p.nscl ? p.kco : 1.0, p.nscl ? 1 : 0);
    }
    ++g_stat[1];
    if (p.nupd == p.nf) {                                           // all updated, some scaled: the substep without scaling, then those
        rc = run_substep(p, 1.0, 0);
        for (int f = 0; f < p.nf && rc == TLAB_OK; ++f)
            if (p.scl[f]) rc = tlab_pw_scale(p.x[f], p.kco, p.n);
        return rc;
    }
    rc = run_rhs(p);
    for (int f = 0; f < p.nf && rc == TLAB_OK; ++f)
        if (p.upd[f]) rc = tlab_pw_rk_update(p.y[f], p.x[f], p.dte, 1.0, 0, p.n);
    return rc;
}
}      // namespace

// capi.cpp: tlab_current_stream(), tlab_sync, the copies, tlab_free, tlab_set_stream
// A recorded substep that runs because some OTHER entry point wanted the stream (the hook in tlab_current_stream()) has no caller to report a failure to:
// the code is kept and handed to the next tlab_sync / tlab_deferred_* call (the error text stays in tlab_last_error()).
static int g_sticky = TLAB_OK;
int tlab_internal_deferred_flush() {
    if (g_busy) return TLAB_OK;
    int rc = TLAB_OK;
    if (g_on && (g_p.rhs || !g_p.zeros.empty())) rc = flush_impl();
    if (rc != TLAB_OK && g_sticky == TLAB_OK) g_sticky = rc;
    return rc;
}
int tlab_internal_deferred_take_error() {      // capi.cpp: tlab_sync
    const int rc = g_sticky;
    g_sticky = TLAB_OK;
    return rc;
}

extern "C" {

int tlab_deferred_enable(int on) {
    const int rc = tlab_internal_deferred_flush();
    g_on = on != 0;
    return rc;
}

int tlab_deferred_flush(void) {
    const int rc = tlab_internal_deferred_flush();
    const int old = tlab_internal_deferred_take_error();
    return rc != TLAB_OK ? rc : old;
}

int tlab_deferred_stats(long long *counts) {
    if (!counts) return TLAB_EINVAL;
    for (int i = 0; i < 6; ++i) counts[i] = g_stat[i];
    return TLAB_OK;
}

int tlab_deferred_zero(double *a, long long n) {
    if (!a || n < 0) { tlab_set_error("tlab_deferred_zero: bad arguments"); return TLAB_EINVAL; }
    if (!g_on) { ++g_stat[5]; return tlab_pw_fill(a, 0.0, n); }
    if (g_p.rhs) {                       // a substep is still described: it runs first (its fields may be the ones zeroed here)
        const int rc = flush_impl();
        if (rc != TLAB_OK) return rc;
    }
    g_p.zeros.push_back({a, n});
    return TLAB_OK;
}

int tlab_deferred_rhs(tlab_dns_t d, double dte, double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc) {
    if (!g_on) return tlab_rhs_global_incompressible_1(d, dte, q, s, hq, hs, txc);
    if (!d || !q || !hq || !txc || dte <= 0.0) { tlab_set_error("tlab_deferred_rhs: bad arguments"); return TLAB_EINVAL; }
    const int ns = tlab_internal_dns_nscal(d);
    if (ns > 0 && (!s || !hs)) { tlab_set_error("tlab_deferred_rhs: bad arguments"); return TLAB_EINVAL; }
    if (g_p.rhs) {
        const int rc = flush_impl();
        if (rc != TLAB_OK) return rc;
    }
    Pending &p = g_p;                    // (keeps the zero fills recorded so far)
    p.rhs = true;
    p.kind = 0; p.d = d; p.slab = nullptr; p.pencil = nullptr;
    p.dte = dte; p.kco = 1.0;
    p.nf = 3 + ns; p.n = tlab_internal_dns_points(d);
    p.q.assign(q, q + 3); p.hq.assign(hq, hq + 3);
    p.s.clear(); p.hs.clear();
    if (ns > 0) { p.s.assign(s, s + ns); p.hs.assign(hs, hs + ns); }
    p.txc.assign(txc, txc + 9);
    p.x.clear(); p.y.clear();
    for (int i = 0; i < 3; ++i) { p.x.push_back(hq[i]); p.y.push_back(q[i]); }
    for (int i = 0; i < ns; ++i) { p.x.push_back(hs[i]); p.y.push_back(s[i]); }
    p.upd.assign(p.nf, 0); p.scl.assign(p.nf, 0);
    p.nupd = p.nscl = 0;
    return TLAB_OK;
}

// the same for the decomposed drivers: their arrays are bound (tlab_slab_dns_bind / tlab_pencil_dns_bind), so the call carries the handle and dte only
static int deferred_decomposed(int kind, tlab_slab_dns_t slab, tlab_pencil_dns_t pencil, double dte) {
    double *const *q = nullptr, *const *s = nullptr, *const *hq = nullptr, *const *hs = nullptr;
    int ns = 0;
    long long n = 0;
    const bool okb = kind == 1 ? tlab_internal_slab_bound(slab, &q, &s, &hq, &hs, &ns, &n) : tlab_internal_pencil_bound(pencil, &q, &s, &hq, &hs, &ns, &n);
    if (!g_on || !okb || !(dte > 0.0)) {      // off, or nothing to match the BLAS calls against (several local ranks): at once
        if (g_on) { const int rc = flush_impl(); if (rc != TLAB_OK) return rc; }
        return kind == 1 ? tlab_slab_dns_rhs(slab, dte) : tlab_pencil_dns_rhs(pencil, dte);
    }
    if (g_p.rhs) {
        const int rc = flush_impl();
        if (rc != TLAB_OK) return rc;
    }
    Pending &p = g_p;
    p.rhs = true;
    p.kind = kind; p.d = nullptr; p.slab = slab; p.pencil = pencil;
    p.dte = dte; p.kco = 1.0;
    p.nf = 3 + ns; p.n = n;
    p.q.clear(); p.s.clear(); p.hq.clear(); p.hs.clear(); p.txc.clear();
    p.x.clear(); p.y.clear();
    for (int i = 0; i < 3; ++i) { p.x.push_back(hq[i]); p.y.push_back(q[i]); }
    for (int i = 0; i < ns; ++i) { p.x.push_back(hs[i]); p.y.push_back(s[i]); }
    p.upd.assign(p.nf, 0); p.scl.assign(p.nf, 0);
    p.nupd = p.nscl = 0;
    return TLAB_OK;
}
int tlab_deferred_slab_rhs(tlab_slab_dns_t d, double dte) {
    if (!d) { tlab_set_error("tlab_deferred_slab_rhs: null handle"); return TLAB_EINVAL; }
    return deferred_decomposed(1, d, nullptr, dte);
}
int tlab_deferred_pencil_rhs(tlab_pencil_dns_t d, double dte) {
    if (!d) { tlab_set_error("tlab_deferred_pencil_rhs: null handle"); return TLAB_EINVAL; }
    return deferred_decomposed(2, nullptr, d, dte);
}

int tlab_deferred_axpy(long long n, double a, const double *x, double *y) {
    if (!x || !y || n < 0) { tlab_set_error("tlab_deferred_axpy: bad arguments"); return TLAB_EINVAL; }
    if (g_on && g_p.rhs && g_p.nscl == 0 && n == g_p.n && a == g_p.dte) {
        for (int f = 0; f < g_p.nf; ++f)
            if (!g_p.upd[f] && g_p.x[f] == x && g_p.y[f] == y) { g_p.upd[f] = 1; ++g_p.nupd; return TLAB_OK; }
    }
    if (g_on) {
        const int rc = flush_impl();
        if (rc != TLAB_OK) return rc;
    }
    ++g_stat[3];
    return tlab_pw_rk_update(y, const_cast<double *>(x), a, 1.0, 0, n);       // y += a x (x is only read when scale = 0)
}

int tlab_deferred_scal(long long n, double a, double *x) {
    if (!x || n < 0) { tlab_set_error("tlab_deferred_scal: bad arguments"); return TLAB_EINVAL; }
    if (g_on && g_p.rhs && g_p.nupd == g_p.nf && n == g_p.n && (g_p.nscl == 0 || a == g_p.kco)) {
        for (int f = 0; f < g_p.nf; ++f)
            if (!g_p.scl[f] && g_p.x[f] == x) {
                g_p.scl[f] = 1; ++g_p.nscl; g_p.kco = a;
                return g_p.nscl == g_p.nf ? flush_impl() : TLAB_OK;      // complete: nothing more to wait for
            }
    }
    if (g_on) {
        const int rc = flush_impl();
        if (rc != TLAB_OK) return rc;
    }
    ++g_stat[4];
    return tlab_pw_scale(x, a, n);
}

}      // extern "C"
