// FDM_Int1 with 3- and 7-diagonal integral systems: the first-order integral operators u' + lambda u = f of the factorized Poisson
// solver when the y plan's first derivative is NOT the tridiagonal / pentadiagonal CompactJacobian6 --
//   SpaceOrder1 = CompactJacobian4 | CompactDirect4   nb_diag = (3, 3)  ->  tridiagonal systems   TRIDFS / TRIDSS,     right-hand side MatMul_3d
//   SpaceOrder1 = CompactJacobian6Penta               nb_diag = (5, 7)  ->  heptadiagonal systems HEPTADFS / HEPTADSS, right-hand side MatMul_5d
// (fdm/fdm_integral.f90:58-87 FDM_Int1_Initialize, :91-214 FDM_Int1_CreateSystem; fdm/fdm_base.f90:304-391 FDM_Bcs_Reduce; utils/linear3.f90:29-51,
// utils/linear7.f90:30-93).  No example selects these schemes together with the factorized solver: correctness first.  The systems of ALL modes are
// built and factorized HERE, on the host, at plan creation, operation by operation as the reference does it (no fused multiply-adds: the solution of
// the Neumann problem is sensitive to the last bit of the factors), and the device kernel (poisson.hip: k_int1g) only substitutes.
// Memory: (nd + 2) x n doubles per mode and system -- 4.2 GB for the 7-diagonal systems of a 512^3 box, of the 288 GB.
#include "int1_generic.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <thread>

#pragma STDC FP_CONTRACT OFF

namespace tlab {

namespace {

// one mode: small dense work arrays in the reference's shapes (1-based accessors as in the Fortran)
struct Mode {
    int nx, ndl, ndr, idl, idr;         // ndl / ndr: diagonals of the DERIVATIVE's lhs (A) and rhs (B); the integral system has ndr, its rhs ndl
    std::vector<double> lhs;            // (nx, ndr)  C = B + lambda A
    std::vector<double> rhs;            // (nx, ndl)  A, reduced + normalised
    double rhs_b[5 * 8], rhs_t[5 * 8];  // rhs_b(1:5, 0:7) [ (j-1) + 5 c ], rhs_t(0:4, 1:8) [ r + 5 (c-1) ]
    double &L(int i, int k) { return lhs[(size_t)(i - 1) + (size_t)nx * (k - 1)]; }
    double &R(int i, int k) { return rhs[(size_t)(i - 1) + (size_t)nx * (k - 1)]; }
    double &RB(int j, int c) { return rhs_b[(j - 1) + 5 * c]; }
    double &RT(int r, int c) { return rhs_t[r + 5 * (c - 1)]; }
};

// FDM_Bcs_Reduce(ibc, lhs, rhs, rhs_b, rhs_t)   fdm_base.f90:304-391.  lhs (nx, nl) modified in place; rhs (nx, nr) read; rb / rt (5 x 8) written
// when given.  ibc: 1 BCS_MIN, 2 BCS_MAX.
void bcs_reduce(int ibc, int nx, double *lhs, int nl, const double *rhs, int nr, double *rb, double *rt) {
    const int idl = nl / 2 + 1, idr = nr / 2 + 1, nx_t = idr, mx = std::max(idl, idr + 1);
    auto L = [&](int i, int k) -> double & { return lhs[(size_t)(i - 1) + (size_t)nx * (k - 1)]; };
    auto Rr = [&](int i, int k) -> double { return rhs[(size_t)(i - 1) + (size_t)nx * (k - 1)]; };
    if (ibc == 1) {
        const double dummy = 1.0 / L(1, idl);
        for (int k = 1; k <= nl; ++k) L(1, k) = -L(1, k) * dummy;
        L(1, idl) = 1.0;
        for (int ir = 1; ir <= idl - 1; ++ir) {
            for (int ic = idl + 1; ic <= nl; ++ic) L(1 + ir, ic - ir) = L(1 + ir, ic - ir) + L(1 + ir, idl - ir) * L(1, ic);
            const int ic = nl + 1;      // longer stencil at the boundary
            L(1 + ir, ic - ir) = L(1 + ir, ic - ir) + L(1 + ir, idl - ir) * L(1, 1);
        }
        if (rb) {
            auto RB = [&](int j, int c) -> double & { return rb[(j - 1) + 5 * c]; };
            for (int j = 1; j <= mx; ++j)
                for (int c = 1; c <= nr; ++c) RB(j, c) = Rr(j, c);
            for (int c = 1; c <= nr; ++c) RB(1, c) = RB(1, c) * dummy;
            for (int ir = 1; ir <= idl - 1; ++ir) {
                for (int ic = idr; ic <= nr; ++ic) RB(1 + ir, ic - ir) = RB(1 + ir, ic - ir) - L(1 + ir, idl - ir) * RB(1, ic);      // ic = idr: b^R_{21}
                const int ic = nr + 1;
                RB(1 + ir, ic - ir) = RB(1 + ir, ic - ir) - L(1 + ir, idl - ir) * RB(1, 1);
            }
        }
    } else {
        const double dummy = 1.0 / L(nx, idl);
        for (int k = 1; k <= nl; ++k) L(nx, k) = -L(nx, k) * dummy;
        L(nx, idl) = 1.0;
        for (int ir = 1; ir <= idl - 1; ++ir) {
            L(nx - ir, ir) = L(nx - ir, ir) + L(nx - ir, idl + ir) * L(nx, nl);      // ic = 0: longer stencil at the boundary
            for (int ic = 1; ic <= idl - 1; ++ic) L(nx - ir, ic + ir) = L(nx - ir, ic + ir) + L(nx - ir, idl + ir) * L(nx, ic);
        }
        if (rt) {
            auto RT = [&](int r, int c) -> double & { return rt[r + 5 * (c - 1)]; };
            for (int j = 0; j < mx; ++j)
                for (int c = 1; c <= nr; ++c) RT(nx_t - mx + 1 + j, c) = Rr(nx - mx + 1 + j, c);
            for (int c = 1; c <= nr; ++c) RT(nx_t, c) = RT(nx_t, c) * dummy;
            for (int ir = 1; ir <= idl - 1; ++ir) {
                RT(nx_t - ir, ir) = RT(nx_t - ir, ir) - L(nx - ir, idl + ir) * RT(nx_t, nr);
                for (int ic = 1; ic <= idr; ++ic) RT(nx_t - ir, ic + ir) = RT(nx_t - ir, ic + ir) - L(nx - ir, idl + ir) * RT(nx_t, ic);
            }
        }
    }
}

// TRIDFS(nmax, a, b, c)   utils/linear3.f90:29-51 (column stride = 1 here: the columns of one mode)
void tridfs(int nmax, double *a, double *b, double *c) {
    for (int n = 1; n < nmax; ++n) {
        a[n] = a[n] / b[n - 1];
        b[n] = b[n] - a[n] * c[n - 1];
    }
    for (int n = 0; n < nmax; ++n) { a[n] = -a[n]; b[n] = 1.0 / b[n]; c[n] = -c[n]; }
}

// HEPTADFS(nmax, a, b, c, d, e, f, g)   utils/linear7.f90:30-93
void heptadfs(int nmax, double *a, double *b, double *c, double *d, double *e, double *f, double *g) {
    g[0] = g[0] / d[0];
    f[0] = f[0] / d[0];
    e[0] = e[0] / d[0];
    c[0] = 1.0 / d[0];
    d[0] = 1.0;
    c[1] = c[1] / d[0];
    d[1] = d[1] - c[1] * e[0];
    e[1] = e[1] - c[1] * f[0];
    f[1] = f[1] - c[1] * g[0];
    b[2] = b[2] / d[0];
    c[2] = (c[2] - b[2] * e[0]) / d[1];
    d[2] = d[2] - c[2] * e[1] - b[2] * f[0];
    e[2] = e[2] - c[2] * f[1] - b[2] * g[0];
    f[2] = f[2] - c[2] * g[1];
    for (int n = 3; n < nmax; ++n) {
        a[n] = a[n] / d[n - 3];
        b[n] = (b[n] - a[n] * e[n - 3]) / d[n - 2];
        c[n] = (c[n] - b[n] * e[n - 2] - a[n] * f[n - 3]) / d[n - 1];
        d[n] = d[n] - c[n] * e[n - 1] - b[n] * f[n - 2] - a[n] * g[n - 3];
        if (n <= nmax - 2) e[n] = e[n] - c[n] * f[n - 1] - b[n] * g[n - 2];
        if (n <= nmax - 3) f[n] = f[n] - c[n] * g[n - 1];
    }
}

// FDM_Int1_CreateSystem + the LU of FDM_Int1_Initialize for one lambda
void build_mode(const DerTables &g, int ibc, double lambda, Mode &m) {
    const int nx = g.n, ndl = g.ndl, ndr = g.ndr, idl = ndl / 2 + 1, idr = ndr / 2 + 1;
    m.nx = nx; m.ndl = ndl; m.ndr = ndr; m.idl = idl; m.idr = idr;
    auto GL = [&](int i, int k) { return g.lhs[(size_t)(i - 1) + (size_t)nx * (k - 1)]; };
    auto GR = [&](int i, int k) { return g.rhs[(size_t)(i - 1) + (size_t)nx * (k - 1)]; };
    m.rhs.assign((size_t)nx * ndl, 0.0);
    for (int k = 1; k <= ndl; ++k)
        for (int i = 1; i <= nx; ++i) m.R(i, k) = GL(i, k);                              // fdmi%rhs = g%lhs (:126)
    double rhsr_b[5 * 8] = {0}, rhsr_t[5 * 8] = {0};
    std::vector<double> grhs((size_t)nx * ndr);
    for (int k = 1; k <= ndr; ++k)
        for (int i = 1; i <= nx; ++i) grhs[(size_t)(i - 1) + (size_t)nx * (k - 1)] = GR(i, k);
    bcs_reduce(ibc, nx, m.rhs.data(), ndl, grhs.data(), ndr, rhsr_b, rhsr_t);             // :128
    auto RRB = [&](int j, int c) { return rhsr_b[(j - 1) + 5 * c]; };
    auto RRT = [&](int r, int c) { return rhsr_t[r + 5 * (c - 1)]; };
    std::fill(m.rhs_b, m.rhs_b + 40, 0.0);
    std::fill(m.rhs_t, m.rhs_t + 40, 0.0);
    if (ibc == 1) {                                                                       // :133-138
        for (int j = 1; j <= idl + 1; ++j)
            for (int c = 1; c <= ndl; ++c) m.RB(j, c) = m.R(j, c);
        for (int ir = 1; ir <= idr - 1; ++ir) m.RB(1 + ir, idl - ir) = -RRB(1 + ir, idr - ir);
    } else {                                                                              // :140-144
        for (int r = 0; r <= idl; ++r)
            for (int c = 1; c <= ndl; ++c) m.RT(r, c) = m.R(nx - idl + r, c);
        for (int ir = 1; ir <= idr - 1; ++ir) m.RT(idl - ir, idl + ir) = -RRT(idr - ir, idr + ir);
    }
    // new lhs diagonals C = B + lambda A (:150-156)
    m.lhs.assign((size_t)nx * ndr, 0.0);
    for (int k = 1; k <= ndr; ++k)
        for (int i = 1; i <= nx; ++i) m.L(i, k) = GR(i, k);
    for (int i = 1; i <= nx; ++i) m.L(i, idr) = m.L(i, idr) + lambda * GL(i, idl);
    for (int i = 1; i <= idl - 1; ++i) {
        for (int r = 1 + i; r <= nx; ++r) m.L(r, idr - i) = m.L(r, idr - i) + lambda * GL(r, idl - i);
        for (int r = 1; r <= nx - i; ++r) m.L(r, idr + i) = m.L(r, idr + i) + lambda * GL(r, idl + i);
    }
    if (ibc == 1) {                                                                       // :159-165
        for (int j = 1; j <= idr; ++j)
            for (int c = 1; c <= ndr; ++c) m.L(j, c) = RRB(j, c);
        for (int c = 1; c <= idl - 1; ++c) m.L(1, idr + c) = m.L(1, idr + c) - lambda * m.RB(1, idl + c);
        for (int ir = 1; ir <= idr - 1; ++ir)
            for (int c = 1; c <= ndl; ++c) m.L(1 + ir, idr - idl + c) = m.L(1 + ir, idr - idl + c) + lambda * m.RB(1 + ir, c);
    } else {                                                                              // :166-172
        for (int j = 1; j <= idr; ++j)
            for (int c = 1; c <= ndr; ++c) m.L(nx - idr + j, c) = RRT(j, c);
        for (int c = 1; c <= idl - 1; ++c) m.L(nx, idr - idl + c) = m.L(nx, idr - idl + c) - lambda * m.RT(idl, c);
        for (int ir = 1; ir <= idr - 1; ++ir)
            for (int c = 1; c <= ndl; ++c) m.L(nx - ir, idr - idl + c) = m.L(nx - ir, idr - idl + c) + lambda * m.RT(idl - ir, c);
    }
    // normalisation such that the new central diagonal of rhs is 1 at the ends (:177-191), the first upper diagonal inside (:194-200)
    const int mx = std::max(idr, idl + 1);
    for (int ir = 1; ir <= mx; ++ir) {
        double dummy = 1.0 / m.R(ir, idl);
        for (int c = 0; c <= ndl; ++c) m.RB(ir, c) = m.RB(ir, c) * dummy;
        dummy = 1.0 / m.R(nx - ir + 1, idl);
        for (int c = 1; c <= ndl + 1; ++c) m.RT(idl - ir + 1, c) = m.RT(idl - ir + 1, c) * dummy;
        dummy = 1.0 / m.R(ir, idl);
        for (int c = 1; c <= ndl; ++c) m.R(ir, c) = m.R(ir, c) * dummy;
        for (int c = 1; c <= ndr; ++c) m.L(ir, c) = m.L(ir, c) * dummy;
        dummy = 1.0 / m.R(nx - ir + 1, idl);
        for (int c = 1; c <= ndl; ++c) m.R(nx - ir + 1, c) = m.R(nx - ir + 1, c) * dummy;
        for (int c = 1; c <= ndr; ++c) m.L(nx - ir + 1, c) = m.L(nx - ir + 1, c) * dummy;
    }
    for (int ir = mx + 1; ir <= nx - mx; ++ir) {
        const double dummy = 1.0 / m.R(ir, idl + 1);
        for (int c = 1; c <= ndl; ++c) m.R(ir, c) = m.R(ir, c) * dummy;
        for (int c = 1; c <= ndr; ++c) m.L(ir, c) = m.L(ir, c) * dummy;
    }
    // reducing the system at the opposite end (:205-210)
    if (ibc == 1) bcs_reduce(2, nx, m.lhs.data(), ndr, m.rhs.data(), ndl, nullptr, m.rhs_t);
    else bcs_reduce(1, nx, m.lhs.data(), ndr, m.rhs.data(), ndl, m.rhs_b, nullptr);
    // LU decomposition of rows 2 .. nx-1 (:71-83)
    double *col[7];
    for (int k = 0; k < ndr; ++k) col[k] = m.lhs.data() + (size_t)nx * k + 1;
    if (ndr == 3) tridfs(nx - 2, col[0], col[1], col[2]);
    else if (ndr == 7) heptadfs(nx - 2, col[0], col[1], col[2], col[3], col[4], col[5], col[6]);
    else throw std::runtime_error("int1_generic: 3 or 7 diagonals (the pentadiagonal systems have their own path)");
}

}  // namespace

bool int1_generic_applies(const DerTables &g) { return (g.ndl == 3 && g.ndr == 3) || (g.ndl == 5 && g.ndr == 7); }

void int1_generic_build(const DerTables &g, int ibc, const double *lam, long long nm, double lam_sign, Int1Gen &out) {
    if (!int1_generic_applies(g)) throw std::runtime_error("int1_generic: first derivative with (3, 3) or (5, 7) diagonals expected");
    if (g.periodic) throw std::runtime_error("Poisson: the wall-normal direction must not be periodic");
    const int nx = g.n, ndi = g.ndr, nri = g.ndl;
    if (nx < 12) throw std::runtime_error("Poisson: too few points in y");
    out.n = nx; out.ndi = ndi; out.nri = nri; out.nm = nm; out.bc = ibc;
    out.fac.assign((size_t)ndi * nx * nm, 0.0);
    out.rb.assign((size_t)40 * nm, 0.0);
    out.rt.assign((size_t)40 * nm, 0.0);
    out.R.assign((size_t)nx * nri, 0.0);
    const unsigned nt = (unsigned)std::max<long long>(1, std::min<long long>(std::min<long long>(std::thread::hardware_concurrency(), 32), nm / 64));
    auto work = [&](long long t0, long long t1) {
        Mode m;
        for (long long t = t0; t < t1; ++t) {
            build_mode(g, ibc, lam_sign * lam[t], m);
            for (int k = 0; k < ndi; ++k)
                for (int j = 0; j < nx; ++j) out.fac[((size_t)k * nx + j) * nm + t] = m.lhs[(size_t)j + (size_t)nx * k];
            for (int q = 0; q < 40; ++q) { out.rb[(size_t)q * nm + t] = m.rhs_b[q]; out.rt[(size_t)q * nm + t] = m.rhs_t[q]; }
            if (t == 0)      // the right-hand side of the integral (A, reduced + normalised) does not depend on lambda
                for (int j = 0; j < nx; ++j)
                    for (int k = 0; k < nri; ++k) out.R[(size_t)j * nri + k] = m.rhs[(size_t)j + (size_t)nx * k];
        }
    };
    if (nt <= 1) {
        work(0, nm);
    } else {
        std::vector<std::thread> th;
        const long long per = (nm + nt - 1) / nt;
        for (unsigned i = 0; i < nt; ++i) th.emplace_back(work, std::min<long long>(nm, i * per), std::min<long long>(nm, (i + 1) * per));
        for (auto &x : th) x.join();
    }
}

}  // namespace tlab
