// Host-side construction of the compact FDM plans (init-time only; the hot path is in kernels.hip).
// See fdm_schemes.hpp for the reference files restated here.
#include "fdm_schemes.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace tlab {

static const double PI_WP = 3.14159265358979323846;  // base/tlab_constants.f90:55

DerTables::DerTables() {
    std::memset(rhs_b, 0, sizeof(rhs_b));
    std::memset(rhs_t, 0, sizeof(rhs_t));
}

// ------------------------------------------------------------------------------------------------
// utils/linear3.f90
// ------------------------------------------------------------------------------------------------
void tridfs(int nmax, double *a, double *b, double *c) {  // :29-51
    for (int n = 1; n < nmax; ++n) {
        a[n] = a[n] / b[n - 1];
        b[n] = b[n] - a[n] * c[n - 1];
    }
    for (int n = 0; n < nmax; ++n) {
        a[n] = -a[n];
        b[n] = 1.0 / b[n];
        c[n] = -c[n];
    }
}

void tridpfs(int nmax, double *a, double *b, double *c, double *d, double *e) {  // :269-316
    c[0] = c[0] / b[0];
    e[0] = a[0] / b[0];
    d[0] = c[nmax - 1];
    for (int n = 1; n < nmax - 2; ++n) {
        b[n] = b[n] - a[n] * c[n - 1];
        c[n] = c[n] / b[n];
        e[n] = -a[n] * e[n - 1] / b[n];
        d[n] = -d[n - 1] * c[n - 1];
    }
    b[nmax - 2] = b[nmax - 2] - a[nmax - 2] * c[nmax - 3];
    e[nmax - 2] = (c[nmax - 2] - a[nmax - 2] * e[nmax - 3]) / b[nmax - 2];
    d[nmax - 2] = a[nmax - 1] - d[nmax - 3] * c[nmax - 3];
    double sum = 0.0;
    for (int n = 0; n < nmax - 1; ++n) sum = sum + d[n] * e[n];
    b[nmax - 1] = b[nmax - 1] - sum;
    for (int n = 0; n < nmax; ++n) {
        b[n] = 1.0 / b[n];
        a[n] = -a[n] * b[n];
        c[n] = -c[n];
        e[n] = -e[n];
    }
}

void tridss1(int nmax, const double *a, const double *b, const double *c, double *f) {  // :56-150, len = 1
    for (int n = 1; n < nmax; ++n) f[n] = f[n] + a[n] * f[n - 1];
    f[nmax - 1] = f[nmax - 1] * b[nmax - 1];
    for (int n = nmax - 2; n >= 0; --n) f[n] = (f[n] + c[n] * f[n + 1]) * b[n];
}

// ------------------------------------------------------------------------------------------------
// fdm/fdm_com1_jacobian.f90:195-291 and fdm/fdm_com2_jacobian.f90:179-282
// 1-based helper views: L(i,k), R(i,k) with i = 1..nx, k = 1..nd
// ------------------------------------------------------------------------------------------------
// utils/linear5.f90:156-203 (0-based n below = the reference's n - 1)
void pentadfs2(int nmax, double *a, double *b, double *c, double *d, double *e) {
    int n = nmax - 1;
    e[n] = 1.0; d[n] = 1.0;
    n = nmax - 2;
    e[n] = 1.0;
    d[n] = d[n] / c[n + 1];
    c[n] = c[n] - d[n] * b[n + 1];
    b[n] = b[n] - d[n] * a[n + 1];
    for (n = nmax - 3; n >= 2; --n) {
        e[n] = e[n] / c[n + 2];
        d[n] = (d[n] - e[n] * b[n + 2]) / c[n + 1];
        c[n] = c[n] - d[n] * b[n + 1] - e[n] * a[n + 2];
        b[n] = b[n] - d[n] * a[n + 1];
    }
    n = 1;
    e[n] = e[n] / c[n + 2];
    d[n] = (d[n] - e[n] * b[n + 2]) / c[n + 1];
    c[n] = c[n] - d[n] * b[n + 1] - e[n] * a[n + 2];
    b[n] = b[n] - d[n] * a[n + 1];
    a[n] = 1.0;
    n = 0;
    e[n] = e[n] / c[n + 2];
    d[n] = (d[n] - e[n] * b[n + 2]) / c[n + 1];
    c[n] = c[n] - d[n] * b[n + 1] - e[n] * a[n + 2];
    b[n] = 1.0;
    a[n] = 1.0;
}

// utils/linear5.f90:207-244, len = 1
void pentadss2_1(int nmax, const double *a, const double *b, const double *c, const double *d, const double *e, double *f) {
    int n = nmax - 2;
    f[n] = f[n] - f[n + 1] * d[n];
    for (n = nmax - 3; n >= 0; --n) f[n] = f[n] - f[n + 1] * d[n] - f[n + 2] * e[n];
    f[0] = f[0] / c[0];
    f[1] = (f[1] - f[0] * b[1]) / c[1];
    for (n = 2; n < nmax; ++n) f[n] = (f[n] - f[n - 1] * b[n] - f[n - 2] * a[n]) / c[n];
}

// utils/linear5.f90:273-347
void pentadpfs(int nmax, double *a, double *b, double *c, double *d, double *e, double *f, double *g) {
    const double a0 = a[0], b0 = b[0], en = e[nmax - 1], dn = d[nmax - 1];
    b[1] = b[1] - d[nmax - 1];
    c[0] = c[0] - e[nmax - 1];
    c[1] = c[1] - e[nmax - 1];
    c[nmax - 2] = c[nmax - 2] - a[0];
    c[nmax - 1] = c[nmax - 1] - a[0];
    d[nmax - 2] = d[nmax - 2] - b[0];
    a[0] = 0.0; a[1] = 0.0; b[0] = 0.0;
    d[nmax - 1] = 0.0; e[nmax - 1] = 0.0; e[nmax - 2] = 0.0;
    pentadfs2(nmax, a, b, c, d, e);
    a[0] = a0; b[0] = b0; e[nmax - 1] = en; d[nmax - 1] = dn;
    for (int n = 0; n < nmax; ++n) { f[n] = 0.0; g[n] = 0.0; }
    f[0] = 1.0; f[nmax - 2] = 1.0;
    g[1] = 1.0; g[nmax - 1] = 1.0;
    pentadss2_1(nmax, a, b, c, d, e, f);
    pentadss2_1(nmax, a, b, c, d, e, g);
    const double m1 = e[nmax - 1] * f[0] + a[0] * f[nmax - 2] + b[0] * f[nmax - 1] + 1.0;
    const double m2 = e[nmax - 1] * g[0] + a[0] * g[nmax - 2] + b[0] * g[nmax - 1];
    const double m3 = d[nmax - 1] * f[0] + e[nmax - 1] * f[1] + a[0] * f[nmax - 1];
    const double m4 = d[nmax - 1] * g[0] + e[nmax - 1] * g[1] + a[0] * g[nmax - 1] + 1.0;
    if ((m1 * m4 - m2 * m3) < 1e-8) throw std::runtime_error("FDM_CreatePlan. Pendad - matrix M not invertible.");     // :333-336
}

namespace {

inline double cshift(const double *v, int nx, int i /*0-based*/, int s) {  // Fortran cshift(v, s)(i) = v(i+s) circular
    int j = (i + s) % nx;
    if (j < 0) j += nx;
    return v[j];
}

// coef_bcN arrays are passed with their length so that the reference's out-of-range read can be reproduced on purpose
struct Coefs {
    const double *p = nullptr;
    int len = 0;
    double at(int k1 /*1-based*/, double oob) const { return (k1 <= len) ? p[k1 - 1] : oob; }
};

void create_system(bool second, int nx, const double *dx1, const double *dx2, int ndl, int ndr, double *lhs, double *rhs,
                   double *rhs_d1, const double coef_int[5], Coefs bc1, Coefs bc2, Coefs bc3, double oob) {
#define L(i, k) lhs[((i)-1) + (size_t)nx * ((k)-1)]
#define R(i, k) rhs[((i)-1) + (size_t)nx * ((k)-1)]
#define D1(i, k) rhs_d1[((i)-1) + (size_t)nx * ((k)-1)]
    const int idl = ndl / 2 + 1, idr = ndr / 2 + 1;
    const double sgn = second ? 1.0 : -1.0;  // symmetry of the mirrored boundary rows (com1 :242, com2 :228)

    for (int i = 1; i <= nx; ++i) L(i, idl) = 1.0;
    for (int ic = 1; ic <= idl - 1; ++ic)
        for (int i = 1; i <= nx; ++i) {
            L(i, idl - ic) = coef_int[ic - 1];
            L(i, idl + ic) = coef_int[ic - 1];
        }
    for (int i = 1; i <= nx; ++i) R(i, idr) = 0.0;
    for (int ic = 1; ic <= idr - 1; ++ic)
        for (int i = 1; i <= nx; ++i) {
            if (second) {
                R(i, idr) = R(i, idr) - 2.0 * coef_int[ic + 1];
                R(i, idr - ic) = coef_int[ic + 1];
                R(i, idr + ic) = coef_int[ic + 1];
            } else {
                R(i, idr - ic) = -coef_int[ic + 1];
                R(i, idr + ic) = coef_int[ic + 1];
            }
        }

    if (bc1.p) {
        int n = 1;
        for (int k = 1; k <= ndl; ++k) L(n, k) = 0.0;
        L(n, idl) = 1.0;
        if (idl > 1) {
            int icmax = std::min(idl - 1, 2);
            for (int k = 1; k <= icmax; ++k) L(n, idl + k) = bc1.at(k, oob);
        }
        for (int k = 1; k <= ndr; ++k) R(n, k) = 0.0;
        int icmax = std::min(idr, 4);
        for (int k = 0; k < icmax; ++k) R(n, idr + k) = bc1.at(3 + k, oob);
        R(n, 1) = bc1.at(3 + icmax, oob);  // extended rhs stencil; out of range for 7 diagonals (reference defect)
        for (int k = 1; k <= ndl; ++k) L(nx, k) = L(1, ndl + 1 - k);
        for (int k = 1; k <= ndr; ++k) R(nx, k) = sgn * R(1, ndr + 1 - k);
    }
    if (bc2.p) {
        int n = 2;
        if (ndl == 3) {
            L(n, 1) = bc2.at(1, oob); L(n, 2) = 1.0; L(n, 3) = bc2.at(2, oob);
        } else if (ndl == 5 && !second) {
            L(n, 1) = 0.0; L(n, 2) = bc2.at(1, oob); L(n, 3) = 1.0; L(n, 4) = bc2.at(2, oob); L(n, 5) = 0.0;
        }
        for (int k = 1; k <= ndr; ++k) R(n, k) = 0.0;
        int icmax = std::min(idr + 1, 4);
        for (int k = 0; k < icmax; ++k) R(n, idr - 1 + k) = bc2.at(3 + k, oob);
        for (int k = 1; k <= ndl; ++k) L(nx - 1, k) = L(2, ndl + 1 - k);
        for (int k = 1; k <= ndr; ++k) R(nx - 1, k) = sgn * R(2, ndr + 1 - k);
    }
    if (bc3.p) {
        int n = 3;
        if (second && ndl == 3) {
            L(n, 1) = bc3.at(1, oob); L(n, 2) = 1.0; L(n, 3) = bc3.at(2, oob);
        } else if (!second && ndl == 5) {
            L(n, 1) = 0.0; L(n, 2) = bc3.at(1, oob); L(n, 3) = 1.0; L(n, 4) = bc3.at(2, oob); L(n, 5) = 0.0;
        }
        for (int k = 1; k <= ndr; ++k) R(n, k) = 0.0;
        int icmax = std::min(idr + 2, 6);
        for (int k = 0; k < icmax; ++k) R(n, idr - 2 + k) = bc3.at(3 + k, oob);
        for (int k = 1; k <= ndl; ++k) L(nx - 2, k) = L(3, ndl + 1 - k);
        for (int k = 1; k <= ndr; ++k) R(nx - 2, k) = sgn * R(3, ndr + 1 - k);
    }

    if (second) {  // com2 :263-274
        for (int i = 1; i <= nx; ++i) D1(i, idl) = -L(i, idl) * dx2[i - 1];
        for (int ic = 1; ic <= idl - 1; ++ic)
            for (int i = 1; i <= nx; ++i) {
                D1(i, idl - ic) = -L(i, idl - ic) * cshift(dx2, nx, i - 1, -ic);
                D1(i, idl + ic) = -L(i, idl + ic) * cshift(dx2, nx, i - 1, +ic);
            }
        for (int i = 1; i <= nx; ++i) L(i, idl) = L(i, idl) * dx1[i - 1] * dx1[i - 1];
        for (int ic = 1; ic <= idl - 1; ++ic)
            for (int i = 1; i <= nx; ++i) {
                L(i, idl - ic) = L(i, idl - ic) * cshift(dx1, nx, i - 1, -ic) * cshift(dx1, nx, i - 1, -ic);
                L(i, idl + ic) = L(i, idl + ic) * cshift(dx1, nx, i - 1, +ic) * cshift(dx1, nx, i - 1, +ic);
            }
    } else {  // com1 :279-284
        for (int i = 1; i <= nx; ++i) L(i, idl) = L(i, idl) * dx1[i - 1];
        for (int ic = 1; ic <= idl - 1; ++ic)
            for (int i = 1; i <= nx; ++i) {
                L(i, idl - ic) = L(i, idl - ic) * cshift(dx1, nx, i - 1, -ic);
                L(i, idl + ic) = L(i, idl + ic) * cshift(dx1, nx, i - 1, +ic);
            }
    }
    for (int k = 1; k <= ndl; ++k)
        for (int i = 1; i <= nx; ++i) L(i, k) = L(i, k) / coef_int[2];
    for (int k = 1; k <= ndr; ++k)
        for (int i = 1; i <= nx; ++i) R(i, k) = R(i, k) / coef_int[2];
    if (second)
        for (int k = 1; k <= ndl; ++k)
            for (int i = 1; i <= nx; ++i) D1(i, k) = D1(i, k) / coef_int[2];
#undef L
#undef R
#undef D1
}

void wavenumbers(int nx, std::vector<double> &wn) {  // fdm_derivative.f90:198-204
    wn.resize(nx);
    for (int i = 1; i <= nx; ++i) {
        if (i <= nx / 2 + 1)
            wn[i - 1] = 2.0 * PI_WP * double(i - 1) / double(nx);
        else
            wn[i - 1] = 2.0 * PI_WP * double(i - 1 - nx) / double(nx);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// fdm/fdm_base.f90:194-300 FDM_Bcs_Neumann
// ------------------------------------------------------------------------------------------------
void fdm_bcs_neumann(int ibc, int nx, int ndl, double *lhs, int ndr, const double *rhs, double *rhs_b, double *rhs_t) {
#define L(i, k) lhs[((i)-1) + (size_t)nx * ((k)-1)]
#define R(i, k) rhs[((i)-1) + (size_t)nx * ((k)-1)]
#define RB(j, c) rhs_b[((j)-1) + 4 * (c)]  /* rhs_b(1:4, 0:7) */
#define RT(r, c) rhs_t[(r) + 5 * ((c)-1)]  /* rhs_t(0:4, 1:7) */
    const int idl = ndl / 2 + 1, idr = ndr / 2 + 1;
    if (ibc == BCS_ND || ibc == BCS_NN) {
        for (int j = 1; j <= idr; ++j)
            for (int c = 1; c <= ndr; ++c) RB(j, c) = R(j, c);
        double dummy = 1.0 / R(1, idr);
        for (int c = 1; c <= ndr; ++c) RB(1, c) = -RB(1, c) * dummy;
        for (int ir = 1; ir <= idr - 1; ++ir) {
            for (int ic = idr + 1; ic <= ndr; ++ic) RB(1 + ir, ic - ir) = RB(1 + ir, ic - ir) + RB(1 + ir, idr - ir) * RB(1, ic);
            int ic = ndr + 1;
            RB(1 + ir, ic - ir) = RB(1 + ir, ic - ir) + RB(1 + ir, idr - ir) * RB(1, 1);
        }
        for (int k = 1; k <= ndl; ++k) L(1, k) = L(1, k) * dummy;
        for (int ir = 1; ir <= idr - 1; ++ir) {
            for (int ic = idl + 1; ic <= ndl; ++ic) L(1 + ir, ic - ir) = L(1 + ir, ic - ir) - RB(1 + ir, idr - ir) * L(1, ic);
            RB(1 + ir, idr - ir) = RB(1 + ir, idr - ir) * L(1, idl);
        }
        for (int ir = 1; ir <= idl - 1; ++ir) RB(1 + ir, idr - ir) = RB(1 + ir, idr - ir) - L(1 + ir, idl - ir);
        RB(1, idr) = L(1, idl);
    }
    if (ibc == BCS_DN || ibc == BCS_NN) {
        for (int j = 1; j <= idr; ++j)
            for (int c = 1; c <= ndr; ++c) RT(j, c) = R(nx - idr + j, c);
        double dummy = 1.0 / R(nx, idr);
        for (int c = 1; c <= ndr; ++c) RT(idr, c) = -RT(idr, c) * dummy;
        for (int ir = 1; ir <= idr - 1; ++ir) {
            for (int ic = 1; ic <= idr - 1; ++ic) RT(idr - ir, ic + ir) = R(nx - ir, ic + ir) + R(nx - ir, idr + ir) * RT(idr, ic);
            RT(idr - ir, ir) = RT(idr - ir, ir) + R(nx - ir, idr + ir) * RT(idr, ndr);
        }
        for (int k = 1; k <= ndl; ++k) L(nx, k) = L(nx, k) * dummy;
        for (int ir = 1; ir <= idr - 1; ++ir) {
            for (int ic = 1; ic <= idl - 1; ++ic) L(nx - ir, ic + ir) = L(nx - ir, ic + ir) - R(nx - ir, idr + ir) * L(nx, ic);
            RT(idr - ir, idr + ir) = RT(idr - ir, idr + ir) * L(nx, idl);
        }
        for (int ir = 1; ir <= idl - 1; ++ir) RT(idr - ir, idr + ir) = RT(idr - ir, idr + ir) - L(nx - ir, idl + ir);
        RT(idr, idr) = L(nx, idl);
    }
#undef L
#undef R
#undef RB
#undef RT
}

// ------------------------------------------------------------------------------------------------
// fdm/fdm_derivative.f90
// ------------------------------------------------------------------------------------------------
void der1_initialize(DerTables &g, int nx, const double *dx, bool periodic, const int *bcs_cases, int ncases, double penta_bc1_ext) {
    g.n = nx;
    g.periodic = periodic;
    g.lhs.assign((size_t)nx * 5, 0.0);
    g.rhs.assign((size_t)nx * 7, 0.0);
    g.rhs_cols = 7;
    g.mwn.assign(nx, 0.0);
    double coef[5];
    const double bc1[6] = {2.0, 0.0, -2.5, 2.0, 0.5, 0.0};                                      // com1 :71-72,:117-118
    const double bc2[6] = {1.0 / 6.0, 0.5, -5.0 / 9.0, -0.5, 1.0, 1.0 / 18.0};                 // com1 :121-122
    switch (g.mode_fdm) {
    case FDM_COM4_JACOBIAN:  // com1 :38-83
        g.ndl = 3; g.ndr = 3;
        coef[0] = 0.25; coef[1] = 0.0; coef[2] = 0.75; coef[3] = 0.0; coef[4] = 0.0;
        create_system(false, nx, dx, nullptr, 3, 3, g.lhs.data(), g.rhs.data(), nullptr, coef,
                      periodic ? Coefs{} : Coefs{bc1, 6}, Coefs{}, Coefs{}, 0.0);
        break;
    case FDM_COM6_JACOBIAN:
    case FDM_COM6_JACOBIAN_HYPER:  // com1 :86-132
        g.ndl = 3; g.ndr = 5;
        coef[0] = 1.0 / 3.0; coef[1] = 0.0; coef[2] = 7.0 / 9.0; coef[3] = 1.0 / 36.0; coef[4] = 0.0;
        create_system(false, nx, dx, nullptr, 3, 5, g.lhs.data(), g.rhs.data(), nullptr, coef,
                      periodic ? Coefs{} : Coefs{bc1, 6}, periodic ? Coefs{} : Coefs{bc2, 6}, Coefs{}, 0.0);
        break;
    case FDM_COM6_JACOBIAN_PENTA: {  // com1 :136-192: pentadiagonal LHS, 7-diagonal RHS, alpha = 0.56
        g.ndl = 5; g.ndr = 7;
        coef[0] = 0.56;
        coef[1] = 0.4 * (-1.0 / 3.0 + coef[0]);
        coef[2] = 0.5 * (1.0 / 6.0) * (9.0 + coef[0] - 20.0 * coef[1]);
        coef[3] = 0.25 * (1.0 / 15.0) * (-9.0 + 32.0 * coef[0] + 62.0 * coef[1]);
        coef[4] = (1.0 / 6.0) * (1.0 / 10.0) * (1.0 - 3.0 * coef[0] + 12.0 * coef[1]);
        const double bc3[8] = {1.0 / 3.0, 1.0 / 3.0, -1.0 / 36.0, -7.0 / 9.0, 0.0, 7.0 / 9.0, 1.0 / 36.0, 0.0};
        create_system(false, nx, dx, nullptr, 5, 7, g.lhs.data(), g.rhs.data(), nullptr, coef,
                      periodic ? Coefs{} : Coefs{bc1, 6}, periodic ? Coefs{} : Coefs{bc2, 6}, periodic ? Coefs{} : Coefs{bc3, 8}, penta_bc1_ext);
        break;
    }
    default:
        throw std::runtime_error("first-derivative scheme not supported (CompactJacobian4/6/6Penta only)");
    }
    if (periodic) {  // :193-211 (cos(wn) with coef(2) kept as in the reference)
        std::vector<double> wn;
        wavenumbers(nx, wn);
        for (int i = 0; i < nx; ++i)
            g.mwn[i] = 2.0 * (coef[2] * std::sin(wn[i]) + coef[3] * std::sin(2.0 * wn[i]) + coef[4] * std::sin(3.0 * wn[i])) /
                       (1.0 + 2.0 * coef[0] * std::cos(wn[i]) + 2.0 * coef[1] * std::cos(wn[i]));
    }
    der1_factorize(g, bcs_cases, ncases);
}

// LU of the first-derivative system and its Neumann variants (fdm_derivative.f90:78-119); also fills rhs_b / rhs_t (FDM_Bcs_Neumann)
void der1_factorize(DerTables &g, const int *bcs_cases, int ncases) {
    const int nx = g.n;
    if (g.periodic) {
        g.lu_cols = g.ndl + 2;
        g.lu.assign((size_t)nx * g.lu_cols, 0.0);
        std::copy(g.lhs.begin(), g.lhs.begin() + (size_t)nx * g.ndl, g.lu.begin());
        double *p = g.lu.data();
        if (g.ndl == 3) tridpfs(nx, p, p + nx, p + 2 * nx, p + 3 * nx, p + 4 * nx);
        else pentadpfs(nx, p, p + nx, p + 2 * nx, p + 3 * nx, p + 4 * nx, p + 5 * nx, p + 6 * nx);      // :90-91
    } else {
        g.lu_cols = 20;
        g.lu.assign((size_t)nx * 20, 0.0);
        for (int ib = 0; ib < ncases; ++ib) {
            double *blk = g.lu.data() + (size_t)nx * (ib * 5);
            std::copy(g.lhs.begin(), g.lhs.begin() + (size_t)nx * g.ndl, blk);
            fdm_bcs_neumann(bcs_cases[ib], nx, g.ndl, blk, g.ndr, g.rhs.data(), g.rhs_b, g.rhs_t);
            int nmin = 0, nmax = nx;
            if (bcs_cases[ib] == BCS_ND || bcs_cases[ib] == BCS_NN) nmin++;
            if (bcs_cases[ib] == BCS_DN || bcs_cases[ib] == BCS_NN) nmax--;
            if (g.ndl == 3) tridfs(nmax - nmin, blk + nmin, blk + nx + nmin, blk + 2 * nx + nmin);
            else pentadfs2(nmax - nmin, blk + nmin, blk + nx + nmin, blk + 2 * nx + nmin, blk + 3 * nx + nmin, blk + 4 * nx + nmin);      // :112-113
        }
    }
}

void der2_initialize(DerTables &g, int nx, const double *dx2, bool periodic, bool uniform, double hyper_bc1_ext) {
    g.n = nx;
    g.periodic = periodic;
    g.lhs.assign((size_t)nx * 5, 0.0);
    g.rhs.assign((size_t)nx * 12, 0.0);
    g.rhs_cols = 12;
    g.mwn.assign(nx, 0.0);
    double coef[5];
    const double bc1[6] = {11.0, 0.0, 13.0, -27.0, 15.0, -1.0};  // com2 :110-111
    const double bc2[6] = {0.1, 0.1, 1.2, -2.4, 1.2, 0.0};       // com2 :114-115
    const double bc3[8] = {2.0 / 11.0, 2.0 / 11.0, 3.0 / 44.0, 12.0 / 11.0, -51.0 / 22.0, 12.0 / 11.0, 3.0 / 44.0, 0.0};  // :168-169
    const double *dx1 = dx2, *dxx = dx2 + nx;
    std::vector<double> rd1((size_t)nx * 3, 0.0);
    switch (g.mode_fdm) {
    case FDM_COM4_JACOBIAN:  // com2 :40-84
        g.ndl = 3; g.ndr = 5;
        coef[0] = 0.1; coef[1] = 0.0; coef[2] = 1.2; coef[3] = 0.0; coef[4] = 0.0;
        create_system(true, nx, dx1, dxx, 3, 5, g.lhs.data(), g.rhs.data(), rd1.data(), coef,
                      periodic ? Coefs{} : Coefs{bc1, 6}, Coefs{}, Coefs{}, 0.0);
        break;
    case FDM_COM6_JACOBIAN:
    case FDM_COM6_JACOBIAN_PENTA:  // com2 :86-122
        g.ndl = 3; g.ndr = 5;
        coef[0] = 2.0 / 11.0; coef[1] = 0.0; coef[2] = 12.0 / 11.0; coef[3] = 3.0 / 44.0; coef[4] = 0.0;
        create_system(true, nx, dx1, dxx, 3, 5, g.lhs.data(), g.rhs.data(), rd1.data(), coef,
                      periodic ? Coefs{} : Coefs{bc1, 6}, periodic ? Coefs{} : Coefs{bc2, 6}, Coefs{}, 0.0);
        break;
    case FDM_COM6_JACOBIAN_HYPER: {  // com2 :125-176
        g.ndl = 3; g.ndr = 7;
        const double kc = std::pow(PI_WP, 2.0);
        coef[0] = (272.0 - 45.0 * kc) / (416.0 - 90.0 * kc);
        coef[1] = 0.0;
        coef[2] = (48.0 - 135.0 * kc) / (1664.0 - 360.0 * kc);
        coef[3] = (528.0 - 81.0 * kc) / (208.0 - 45.0 * kc) / 4.0;
        coef[4] = -(432.0 - 63.0 * kc) / (1664.0 - 360.0 * kc) / 9.0;
        create_system(true, nx, dx1, dxx, 3, 7, g.lhs.data(), g.rhs.data(), rd1.data(), coef,
                      periodic ? Coefs{} : Coefs{bc1, 6}, periodic ? Coefs{} : Coefs{bc2, 6},
                      periodic ? Coefs{} : Coefs{bc3, 8}, hyper_bc1_ext);
        break;
    }
    default:
        throw std::runtime_error("second-derivative scheme not supported (CompactJacobian4/6/6Hyper only)");
    }
    if (!uniform) g.need_1der = true;  // fdm_derivative.f90:367,371,375
    std::copy(rd1.begin(), rd1.end(), g.rhs.begin() + (size_t)nx * g.ndr);  // rhs(:, ndr+1:ndr+3), :356,:374
    if (periodic) {  // :389-406
        std::vector<double> wn;
        wavenumbers(nx, wn);
        for (int i = 0; i < nx; ++i)
            g.mwn[i] = 2.0 * (coef[2] * (1.0 - std::cos(wn[i])) + coef[3] * (1.0 - std::cos(2.0 * wn[i])) + coef[4] * (1.0 - std::cos(3.0 * wn[i]))) /
                       (1.0 + 2.0 * coef[0] * std::cos(wn[i]) + 2.0 * coef[1] * std::cos(2.0 * wn[i]));
    }
    // LU (:292-314)
    if (periodic) {
        g.lu_cols = g.ndl + 2;
        g.lu.assign((size_t)nx * g.lu_cols, 0.0);
        std::copy(g.lhs.begin(), g.lhs.begin() + (size_t)nx * g.ndl, g.lu.begin());
        double *p = g.lu.data();
        if (g.ndl == 3) tridpfs(nx, p, p + nx, p + 2 * nx, p + 3 * nx, p + 4 * nx);
        else pentadpfs(nx, p, p + nx, p + 2 * nx, p + 3 * nx, p + 4 * nx, p + 5 * nx, p + 6 * nx);      // :90-91
    } else {
        g.lu_cols = g.ndl;
        g.lu.assign((size_t)nx * g.lu_cols, 0.0);
        std::copy(g.lhs.begin(), g.lhs.begin() + (size_t)nx * g.ndl, g.lu.begin());
        double *p = g.lu.data();
        tridfs(nx, p, p + nx, p + 2 * nx);
    }
}

// ------------------------------------------------------------------------------------------------
// fdm/fdm_matmul.f90 for one line (plan creation only: Jacobians of fdm.f90:201,224).  Non-periodic, ibc = DD.
// ------------------------------------------------------------------------------------------------
void der1_matmul1(const DerTables &g, int ibc, const double *u, double *f) {
    (void)ibc;
    const int nx = g.n;
    const double *r = g.rhs.data();
#define RI(i, k) r[((i)-1) + (size_t)nx * ((k)-1)]
#define U(i) u[(i)-1]
#define F(i) f[(i)-1]
    if (g.ndr == 3) {  // MatMul_3d_antisym :157-212
        F(1) = U(1) * RI(1, 2) + U(2) * RI(1, 3) + U(3) * RI(1, 1);
        F(2) = U(1) * RI(2, 1) + U(2) * RI(2, 2) + U(3) * RI(2, 3);
        for (int n = 3; n <= nx - 2; ++n) F(n) = U(n + 1) - U(n - 1);
        F(nx - 1) = U(nx - 2) * RI(nx - 1, 1) + U(nx - 1) * RI(nx - 1, 2) + U(nx) * RI(nx - 1, 3);
        F(nx) = U(nx - 2) * RI(nx, 3) + U(nx - 1) * RI(nx, 1) + U(nx) * RI(nx, 2);
    } else if (g.ndr == 7) {  // MatMul_7d_antisym :491-558 (biased rows)
        const double r6 = RI(5, 6), r7 = RI(5, 7);
        F(1) = U(1) * RI(1, 4) + U(2) * RI(1, 5) + U(3) * RI(1, 6) + U(4) * RI(1, 7) + U(5) * RI(1, 1);
        F(2) = U(1) * RI(2, 3) + U(2) * RI(2, 4) + U(3) * RI(2, 5) + U(4) * RI(2, 6) + U(5) * RI(2, 7);
        F(3) = U(1) * RI(3, 2) + U(2) * RI(3, 3) + U(3) * RI(3, 4) + U(4) * RI(3, 5) + U(5) * RI(3, 6) + U(6) * RI(3, 7);
        F(4) = U(1) * RI(4, 1) + U(2) * RI(4, 2) + U(3) * RI(4, 3) + U(4) * RI(4, 4) + U(5) * RI(4, 5) + U(6) * RI(4, 6) + U(7) * RI(4, 7);
        for (int n = 5; n <= nx - 4; ++n) F(n) = U(n + 1) - U(n - 1) + r6 * (U(n + 2) - U(n - 2)) + r7 * (U(n + 3) - U(n - 3));
        F(nx - 3) = U(nx - 6) * RI(nx - 3, 1) + U(nx - 5) * RI(nx - 3, 2) + U(nx - 4) * RI(nx - 3, 3) + U(nx - 3) * RI(nx - 3, 4) + U(nx - 2) * RI(nx - 3, 5) + U(nx - 1) * RI(nx - 3, 6) + U(nx) * RI(nx - 3, 7);
        F(nx - 2) = U(nx - 5) * RI(nx - 2, 1) + U(nx - 4) * RI(nx - 2, 2) + U(nx - 3) * RI(nx - 2, 3) + U(nx - 2) * RI(nx - 2, 4) + U(nx - 1) * RI(nx - 2, 5) + U(nx) * RI(nx - 2, 6);
        F(nx - 1) = U(nx - 4) * RI(nx - 1, 1) + U(nx - 3) * RI(nx - 1, 2) + U(nx - 2) * RI(nx - 1, 3) + U(nx - 1) * RI(nx - 1, 4) + U(nx) * RI(nx - 1, 5);
        F(nx) = U(nx - 4) * RI(nx, 7) + U(nx - 3) * RI(nx, 1) + U(nx - 2) * RI(nx, 2) + U(nx - 1) * RI(nx, 3) + U(nx) * RI(nx, 4);
    } else {  // MatMul_5d_antisym :359-419
        const double r5 = RI(4, 5);
        F(1) = U(1) * RI(1, 3) + U(2) * RI(1, 4) + U(3) * RI(1, 5) + U(4) * RI(1, 1);
        F(2) = U(1) * RI(2, 2) + U(2) * RI(2, 3) + U(3) * RI(2, 4) + U(4) * RI(2, 5);
        F(3) = U(1) * RI(3, 1) + U(2) * RI(3, 2) + U(3) * RI(3, 3) + U(4) * RI(3, 4) + U(5) * RI(3, 5);
        for (int n = 4; n <= nx - 3; ++n) F(n) = U(n + 1) - U(n - 1) + r5 * (U(n + 2) - U(n - 2));
        F(nx - 2) = U(nx - 4) * RI(nx - 2, 1) + U(nx - 3) * RI(nx - 2, 2) + U(nx - 2) * RI(nx - 2, 3) + U(nx - 1) * RI(nx - 2, 4) + U(nx) * RI(nx - 2, 5);
        F(nx - 1) = U(nx - 3) * RI(nx - 1, 1) + U(nx - 2) * RI(nx - 1, 2) + U(nx - 1) * RI(nx - 1, 3) + U(nx) * RI(nx - 1, 4);
        F(nx) = U(nx - 3) * RI(nx, 5) + U(nx - 2) * RI(nx, 1) + U(nx - 1) * RI(nx, 2) + U(nx) * RI(nx, 3);
    }
}

void der2_matmul1(const DerTables &g, int ibc, const double *u, double *f) {
    (void)ibc;
    const int nx = g.n;
    const double *r = g.rhs.data();
    if (g.ndr == 5) {  // MatMul_5d_sym :423-485
        const double r5 = RI(3, 5), r3 = RI(3, 3);
        F(1) = U(1) * RI(1, 3) + U(2) * RI(1, 4) + U(3) * RI(1, 5) + U(4) * RI(1, 1);
        F(2) = U(1) * RI(2, 2) + U(2) * RI(2, 3) + U(3) * RI(2, 4) + U(4) * RI(2, 5);
        for (int n = 3; n <= nx - 2; ++n) F(n) = r3 * U(n) + U(n + 1) + U(n - 1) + r5 * (U(n + 2) + U(n - 2));
        F(nx - 1) = U(nx - 3) * RI(nx - 1, 1) + U(nx - 2) * RI(nx - 1, 2) + U(nx - 1) * RI(nx - 1, 3) + U(nx) * RI(nx - 1, 4);
        F(nx) = U(nx - 3) * RI(nx, 5) + U(nx - 2) * RI(nx, 1) + U(nx - 1) * RI(nx, 2) + U(nx) * RI(nx, 3);
    } else {  // MatMul_7d_sym :562-642
        const double r7 = RI(4, 7), r6 = RI(4, 6), r4 = RI(4, 4);
        F(1) = U(1) * RI(1, 4) + U(2) * RI(1, 5) + U(3) * RI(1, 6) + U(4) * RI(1, 7) + U(5) * RI(1, 1);
        F(2) = U(1) * RI(2, 3) + U(2) * RI(2, 4) + U(3) * RI(2, 5) + U(4) * RI(2, 6) + U(5) * RI(2, 7);
        F(3) = U(1) * RI(3, 2) + U(2) * RI(3, 3) + U(3) * RI(3, 4) + U(4) * RI(3, 5) + U(5) * RI(3, 6) + U(6) * RI(3, 7);
        for (int n = 4; n <= nx - 3; ++n)
            F(n) = r4 * U(n) + U(n + 1) + U(n - 1) + r6 * (U(n + 2) + U(n - 2)) + r7 * (U(n + 3) + U(n - 3));
        F(nx - 2) = U(nx - 5) * RI(nx - 2, 1) + U(nx - 4) * RI(nx - 2, 2) + U(nx - 3) * RI(nx - 2, 3) + U(nx - 2) * RI(nx - 2, 4) + U(nx - 1) * RI(nx - 2, 5) + U(nx) * RI(nx - 2, 6);
        F(nx - 1) = U(nx - 4) * RI(nx - 1, 1) + U(nx - 3) * RI(nx - 1, 2) + U(nx - 2) * RI(nx - 1, 3) + U(nx - 1) * RI(nx - 1, 4) + U(nx) * RI(nx - 1, 5);
        F(nx) = U(nx - 4) * RI(nx, 7) + U(nx - 3) * RI(nx, 1) + U(nx - 2) * RI(nx, 2) + U(nx - 1) * RI(nx, 3) + U(nx) * RI(nx, 4);
    }
#undef RI
#undef U
#undef F
}

// ------------------------------------------------------------------------------------------------
// fdm/fdm_interpolate.f90:33-96 FDM_Interpol_Initialize
// ------------------------------------------------------------------------------------------------
void interpol_initialize(FdmTables &g, bool replace_mwn) {
    const int nx = g.n;
    if (!g.periodic || nx < 8) throw std::runtime_error("staggered grid only along periodic directions (fdm.f90:239-246)");
    const double *dx = g.jac.data();       // jac(:, 1)
    g.lu0i.assign((size_t)5 * nx, 0.0);
    g.lu1i.assign((size_t)5 * nx, 0.0);
    double *a = g.lu0i.data(), *b = a + nx, *c = b + nx;
    for (int i = 0; i < nx; ++i) { a[i] = 2.0 / 5.0; b[i] = 4.0 / 3.0; c[i] = 2.0 / 5.0; }      // FDM_C0INT6P_LHS
    tridpfs(nx, a, b, c, c + nx, c + 2 * nx);
    a = g.lu1i.data(); b = a + nx; c = b + nx;
    for (int i = 0; i < nx; ++i) { a[i] = 9.0 / 63.0; b[i] = 62.0 / 63.0; c[i] = 9.0 / 63.0; }  // FDM_C1INT6P_LHS
    c[nx - 1] = c[nx - 1] * dx[0]; b[0] = b[0] * dx[0]; a[1] = a[1] * dx[0];                     // Jacobian multiplication (:305-317)
    for (int i = 1; i < nx - 1; ++i) { c[i - 1] = c[i - 1] * dx[i]; b[i] = b[i] * dx[i]; a[i + 1] = a[i + 1] * dx[i]; }
    c[nx - 2] = c[nx - 2] * dx[nx - 1]; b[nx - 1] = b[nx - 1] * dx[nx - 1]; a[0] = a[0] * dx[nx - 1];
    tridpfs(nx, a, b, c, c + nx, c + 2 * nx);
    g.stagger = true;
    if (replace_mwn) {
        std::vector<double> wn;
        wavenumbers(nx, wn);
        const double c1 = 9.0 / 62.0, c3 = 63.0 / 62.0, c4 = 17.0 / 62.0;
        g.der1.mwn.resize(nx);
        for (int i = 0; i < nx; ++i)
            g.der1.mwn[i] = 2.0 * (c3 * std::sin(1.0 / 2.0 * wn[i]) + c4 / 3.0 * std::sin(3.0 / 2.0 * wn[i])) / (1.0 + 2.0 * c1 * std::cos(wn[i])) / dx[0];
    }
}

// ------------------------------------------------------------------------------------------------
// fdm/fdm.f90:143-252 FDM_CreatePlan
// ------------------------------------------------------------------------------------------------
void fdm_create_plan(FdmTables &g, int nx, const double *nodes, bool periodic, bool uniform, int mode1, int mode2,
                     double hyper_bc1_ext, double penta_bc1_ext) {
    if (periodic && mode1 == FDM_COM4_DIRECT) mode1 = FDM_COM4_JACOBIAN;  // :155-158
    if (periodic && mode1 == FDM_COM6_DIRECT) mode1 = FDM_COM6_JACOBIAN;
    if (periodic && mode2 == FDM_COM4_DIRECT) mode2 = FDM_COM4_JACOBIAN;
    if (periodic && mode2 == FDM_COM6_DIRECT) mode2 = FDM_COM6_JACOBIAN_HYPER;
    g.n = nx;
    g.periodic = periodic;
    g.uniform = uniform;
    g.nodes.assign(nodes, nodes + nx);
    g.jac.assign((size_t)nx * 3, 1.0);
    g.der1 = DerTables();
    g.der2 = DerTables();
    g.der1.mode_fdm = mode1;
    g.der2.mode_fdm = mode2;
    if (nx == 1) return;

    // first derivative: Jacobian dx/ds from a unit-grid derivative of the node positions (:194-201)
    std::vector<double> ones(nx, 1.0), tmp(nx);
    const int dd[1] = {BCS_DD};
    der1_initialize(g.der1, nx, ones.data(), false, dd, 1, penta_bc1_ext);
    der1_matmul1(g.der1, BCS_DD, nodes, tmp.data());
    {
        const double *lu = g.der1.lu.data();
        if (g.der1.ndl == 3) tridss1(nx, lu, lu + nx, lu + 2 * nx, tmp.data());
        else pentadss2_1(nx, lu, lu + nx, lu + 2 * nx, lu + 3 * nx, lu + 4 * nx, tmp.data());
    }
    std::copy(tmp.begin(), tmp.end(), g.jac.begin());
    const int all[4] = {BCS_DD, BCS_ND, BCS_DN, BCS_NN};
    der1_initialize(g.der1, nx, g.jac.data(), periodic, all, 4, penta_bc1_ext);  // :207
    if (periodic)
        for (int i = 0; i < nx; ++i) g.der1.mwn[i] = g.der1.mwn[i] / g.jac[0];  // :209

    // second derivative (:212-233)
    std::vector<double> j2((size_t)nx * 2);
    for (int i = 0; i < nx; ++i) { j2[i] = 1.0; j2[nx + i] = 0.0; }
    der2_initialize(g.der2, nx, j2.data(), false, true, hyper_bc1_ext);
    g.der2.need_1der = false;
    der2_matmul1(g.der2, BCS_DD, nodes, tmp.data());
    tridss1(nx, g.der2.lu.data(), g.der2.lu.data() + nx, g.der2.lu.data() + 2 * nx, tmp.data());
    for (int i = 0; i < nx; ++i) {
        g.jac[2 * nx + i] = tmp[i];   // d2x/ds2
        g.jac[nx + i] = g.jac[i];     // :229
    }
    der2_initialize(g.der2, nx, g.jac.data() + nx, periodic, uniform, hyper_bc1_ext);  // :231
    if (periodic)
        for (int i = 0; i < nx; ++i) g.der2.mwn[i] = g.der2.mwn[i] / (g.jac[0] * g.jac[0]);  // :233
}

}  // namespace tlab
