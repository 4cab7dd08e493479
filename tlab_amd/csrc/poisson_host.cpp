// Host (init-time) part of the Poisson solver: lambda-independent tables of the first-order integral operators.
//
// Restates FDM_Int1_CreateSystem (fdm/fdm_integral.f90:91-214) and FDM_Bcs_Reduce (fdm/fdm_base.f90:304-391) for the
// tridiagonal-LHS / pentadiagonal-RHS first-derivative schemes (C1N6: ndl = 3, ndr = 5).  Everything the reference
// computes per Fourier mode that is LINEAR in the mode's constant lambda is kept here as its operands (c0, c1) and the row
// normalisation S: lhs(lambda) = (L0 + lambda * L1) * S is evaluated per mode on the device with the reference's roundings
// (two per entry + one for S), as are the only non-linear step (the reduction of the opposite boundary, :203-211, which
// divides by a lambda-dependent pivot) and the pentadiagonal LU (poisson.hip).
#include "poisson_host.hpp"

#include <algorithm>
#include <stdexcept>

namespace tlab {

namespace {

struct Lin {  // c0 + lambda * c1
    double c0 = 0.0, c1 = 0.0;
};
inline Lin operator*(Lin a, double s) { return Lin{a.c0 * s, a.c1 * s}; }
inline Lin operator-(Lin a, Lin b) { return Lin{a.c0 - b.c0, a.c1 - b.c1}; }

// FDM_Bcs_Reduce for plain doubles (fdm_base.f90:304-391).  1-based accessors.
struct Reduce {
    int nx, ndl, ndr;
    double *lhs;        // (nx, ndl) column-major
    const double *rhs;  // (nx, ndr) column-major, leading dimension nx
    double *rhs_b;      // rhs_b(1:5, 0:7)  -> [(j-1) + 5*c]
    double *rhs_t;      // rhs_t(0:4, 1:8)  -> [r + 5*(c-1)]
    double &L(int i, int k) { return lhs[(i - 1) + (size_t)nx * (k - 1)]; }
    double R(int i, int k) const { return rhs[(i - 1) + (size_t)nx * (k - 1)]; }
    double &RB(int j, int c) { return rhs_b[(j - 1) + 5 * c]; }
    double &RT(int r, int c) { return rhs_t[r + 5 * (c - 1)]; }

    void run(int ibc) {  // ibc: 1 = BCS_MIN, 2 = BCS_MAX
        const int idl = ndl / 2 + 1, idr = ndr / 2 + 1, nx_t = idr, mx = std::max(idl, idr + 1);
        if (ibc == 1) {
            const double dummy = 1.0 / L(1, idl);
            for (int k = 1; k <= ndl; ++k) L(1, k) = -L(1, k) * dummy;
            L(1, idl) = 1.0;
            for (int ir = 1; ir <= idl - 1; ++ir) {
                for (int ic = idl + 1; ic <= ndl; ++ic) L(1 + ir, ic - ir) = L(1 + ir, ic - ir) + L(1 + ir, idl - ir) * L(1, ic);
                const int ic = ndl + 1;
                L(1 + ir, ic - ir) = L(1 + ir, ic - ir) + L(1 + ir, idl - ir) * L(1, 1);
            }
            for (int j = 1; j <= mx; ++j)
                for (int c = 1; c <= ndr; ++c) RB(j, c) = R(j, c);
            for (int c = 1; c <= ndr; ++c) RB(1, c) = RB(1, c) * dummy;
            for (int ir = 1; ir <= idl - 1; ++ir) {
                for (int ic = idr; ic <= ndr; ++ic) RB(1 + ir, ic - ir) = RB(1 + ir, ic - ir) - L(1 + ir, idl - ir) * RB(1, ic);
                const int ic = ndr + 1;
                RB(1 + ir, ic - ir) = RB(1 + ir, ic - ir) - L(1 + ir, idl - ir) * RB(1, 1);
            }
        } else {
            const double dummy = 1.0 / L(nx, idl);
            for (int k = 1; k <= ndl; ++k) L(nx, k) = -L(nx, k) * dummy;
            L(nx, idl) = 1.0;
            for (int ir = 1; ir <= idl - 1; ++ir) {
                L(nx - ir, ir) = L(nx - ir, ir) + L(nx - ir, idl + ir) * L(nx, ndl);
                for (int ic = 1; ic <= idl - 1; ++ic) L(nx - ir, ic + ir) = L(nx - ir, ic + ir) + L(nx - ir, idl + ir) * L(nx, ic);
            }
            for (int j = 0; j < mx; ++j)
                for (int c = 1; c <= ndr; ++c) RT(nx_t - mx + 1 + j, c) = R(nx - mx + 1 + j, c);
            for (int c = 1; c <= ndr; ++c) RT(nx_t, c) = RT(nx_t, c) * dummy;
            for (int ir = 1; ir <= idl - 1; ++ir) {
                RT(nx_t - ir, ir) = RT(nx_t - ir, ir) - L(nx - ir, idl + ir) * RT(nx_t, ndr);
                for (int ic = 1; ic <= idr; ++ic) RT(nx_t - ir, ic + ir) = RT(nx_t - ir, ic + ir) - L(nx - ir, idl + ir) * RT(nx_t, ic);
            }
        }
    }
};

}  // namespace

void int1_build_tables(const DerTables &g, int ibc, Int1Tables &out) {
    if (g.ndl != 3 || g.ndr != 5) throw std::runtime_error("Poisson: needs the CompactJacobian6 first derivative (3/5 diagonals)");
    if (g.periodic) throw std::runtime_error("Poisson: the wall-normal direction must not be periodic");
    const int nx = g.n, ndl = 3, ndr = 5, idl = 2, idr = 3;
    if (nx < 8) throw std::runtime_error("Poisson: too few points in y");
    out.n = nx;
    out.bc = ibc;

    std::vector<double> A(g.lhs.begin(), g.lhs.begin() + (size_t)nx * ndl);  // fdmi%rhs
    double rhsr_b[5 * 8] = {0}, rhsr_t[5 * 8] = {0};
    Reduce red{nx, ndl, ndr, A.data(), g.rhs.data(), rhsr_b, rhsr_t};
    red.run(ibc);
#define AA(i, k) A[((i)-1) + (size_t)nx * ((k)-1)]
#define A0(i, k) g.lhs[((i)-1) + (size_t)nx * ((k)-1)]
#define BB(i, k) g.rhs[((i)-1) + (size_t)nx * ((k)-1)]
#define RRB(j, c) rhsr_b[((j)-1) + 5 * (c)]
#define RRT(r, c) rhsr_t[(r) + 5 * ((c)-1)]
    double rhs_b[5 * 8] = {0}, rhs_t[5 * 8] = {0};
#define RB(j, c) rhs_b[((j)-1) + 5 * (c)]
#define RT(r, c) rhs_t[(r) + 5 * ((c)-1)]
    if (ibc == 1) {  // :134-138
        for (int j = 1; j <= idl + 1; ++j)
            for (int c = 1; c <= ndl; ++c) RB(j, c) = AA(j, c);
        for (int ir = 1; ir <= idr - 1; ++ir) RB(1 + ir, idl - ir) = -RRB(1 + ir, idr - ir);
    } else {  // :140-144
        for (int r = 0; r <= idl; ++r)
            for (int c = 1; c <= ndl; ++c) RT(r, c) = AA(nx - idl + r, c);
        for (int ir = 1; ir <= idr - 1; ++ir) RT(idl - ir, idl + ir) = -RRT(idr - ir, idr + ir);
    }
    // lhs = B + lambda A  (:150-156), kept as (c0, c1)
    std::vector<Lin> lhs((size_t)nx * ndr);
#define LL(i, k) lhs[((i)-1) + (size_t)nx * ((k)-1)]
    for (int k = 1; k <= ndr; ++k)
        for (int i = 1; i <= nx; ++i) LL(i, k) = Lin{BB(i, k), 0.0};
    for (int i = 1; i <= nx; ++i) LL(i, idr).c1 += A0(i, idl);
    for (int ii = 1; ii <= idl - 1; ++ii) {
        for (int i = 1 + ii; i <= nx; ++i) LL(i, idr - ii).c1 += A0(i, idl - ii);
        for (int i = 1; i <= nx - ii; ++i) LL(i, idr + ii).c1 += A0(i, idl + ii);
    }
    if (ibc == 1) {  // :159-165
        for (int j = 1; j <= idr; ++j)
            for (int c = 1; c <= ndr; ++c) LL(j, c) = Lin{RRB(j, c), 0.0};
        for (int c = 0; c < idl - 1; ++c) LL(1, idr + 1 + c).c1 -= RB(1, idl + 1 + c);
        for (int ir = 1; ir <= idr - 1; ++ir)
            for (int c = 0; c < ndl; ++c) LL(1 + ir, idr - idl + 1 + c).c1 += RB(1 + ir, 1 + c);
    } else {  // :166-172
        for (int j = 1; j <= idr; ++j)
            for (int c = 1; c <= ndr; ++c) LL(nx - idr + j, c) = Lin{RRT(j, c), 0.0};
        for (int c = 0; c < idl - 1; ++c) LL(nx, idr - idl + 1 + c).c1 -= RT(idl, 1 + c);
        for (int ir = 1; ir <= idr - 1; ++ir)
            for (int c = 0; c < ndl; ++c) LL(nx - ir, idr - idl + 1 + c).c1 += RT(idl - ir, 1 + c);
    }
    // normalisation (:175-201)
    std::vector<double> S((size_t)nx, 1.0);
    const int mx = std::max(idr, idl + 1);
    for (int ir = 1; ir <= mx; ++ir) {
        double dummy = 1.0 / AA(ir, idl);
        for (int c = 0; c <= ndl; ++c) RB(ir, c) = RB(ir, c) * dummy;
        dummy = 1.0 / AA(nx - ir + 1, idl);
        for (int c = 1; c <= ndl + 1; ++c) RT(idl - ir + 1, c) = RT(idl - ir + 1, c) * dummy;
        dummy = 1.0 / AA(ir, idl);
        for (int c = 1; c <= ndl; ++c) AA(ir, c) = AA(ir, c) * dummy;
        S[ir - 1] = dummy;                           // lhs(ir, :) * dummy is done per mode on the device, after B + lambda A
        dummy = 1.0 / AA(nx - ir + 1, idl);
        for (int c = 1; c <= ndl; ++c) AA(nx - ir + 1, c) = AA(nx - ir + 1, c) * dummy;
        S[nx - ir] = dummy;
    }
    for (int ir = mx + 1; ir <= nx - mx; ++ir) {
        const double dummy = 1.0 / AA(ir, idl + 1);
        for (int c = 1; c <= ndl; ++c) AA(ir, c) = AA(ir, c) * dummy;
        S[ir - 1] = dummy;
    }
    // pack row-major for the device
    out.L0.assign((size_t)nx * 5 + nx, 0.0);         // [n][5] c0, then the row normalisation [n]
    out.L1.assign((size_t)nx * 5, 0.0);
    for (int i = 0; i < nx; ++i) out.L0[(size_t)nx * 5 + i] = S[i];
    out.R.assign((size_t)nx * 3, 0.0);
    for (int i = 1; i <= nx; ++i) {
        for (int k = 1; k <= 5; ++k) {
            out.L0[(size_t)(i - 1) * 5 + (k - 1)] = LL(i, k).c0;
            out.L1[(size_t)(i - 1) * 5 + (k - 1)] = LL(i, k).c1;
        }
        for (int k = 1; k <= 3; ++k) out.R[(size_t)(i - 1) * 3 + (k - 1)] = AA(i, k);
    }
    for (int j = 1; j <= 3; ++j)
        for (int c = 0; c <= 3; ++c) out.rb[j - 1][c] = RB(j, c);
    for (int r = 0; r <= 2; ++r)
        for (int c = 1; c <= 4; ++c) out.rt[r][c - 1] = RT(r, c);
#undef AA
#undef A0
#undef BB
#undef RRB
#undef RRT
#undef RB
#undef RT
#undef LL
}

namespace {

// fdm/fdm_base.f90:31-125 (1-based node and set indices)
struct Poly {
    const std::vector<double> &x;
    double X(int k) const { return x[k - 1]; }
    double Pi(int j, const int (&idx)[3]) const {
        double f = 1.0;
        for (int k = 0; k < 3; ++k) f = f * (X(j) - X(idx[k]));
        return f;
    }
    double Pi_p(int j, const int (&idx)[3]) const {
        double f = 0.0;
        for (int k = 0; k < 3; ++k) {
            double d = 1.0;
            for (int m = 0; m < 3; ++m)
                if (m != k) d = d * (X(j) - X(idx[m]));
            f = f + d;
        }
        return f;
    }
    double Pi_pp_3(int j, const int (&idx)[3]) const { return 2.0 * (X(j) - X(idx[0]) + X(j) - X(idx[1]) + X(j) - X(idx[2])); }
    double Lag(int j, int i, const int (&idx)[3]) const {
        double f = 1.0;
        for (int k = 0; k < 3; ++k)
            if (idx[k] != i) f = f * (X(j) - X(idx[k])) / (X(i) - X(idx[k]));
        return f;
    }
    double Lag_p(int j, int i, const int (&idx)[3]) const {
        double den = 1.0, f = 0.0;
        for (int k = 0; k < 3; ++k) {
            if (idx[k] == i) continue;
            double d = 1.0;
            for (int m = 0; m < 3; ++m)
                if (idx[m] != i && m != k) d = d * (X(j) - X(idx[m]));
            f = f + d;
            den = den * (X(i) - X(idx[k]));
        }
        return f / den;
    }
};

// p'_1 = b1 p1 + b2 p2 + b3 p3 + b4 p4 + a2 p''_2 (coef_c1n4_biased, contained in FDM_Int2_CreateSystem, fdm_integral.f90:560-621)
void coef_c1n4_biased(const std::vector<double> &x, int i, bool backwards, double (&coef)[5]) {
    const Poly P{x};
    const int i1 = i, i2 = backwards ? i - 1 : i + 1, i3 = backwards ? i - 2 : i + 2, i4 = backwards ? i - 3 : i + 3;
    const double dx1 = P.X(i2) - P.X(i1), dx3 = P.X(i2) - P.X(i3), dx4 = P.X(i2) - P.X(i4);
    const int sm[3] = {i1, i3, i4};
    const double a2 = 0.5 * (P.Pi(i1, sm) - dx1 * P.Pi_p(i1, sm)) / P.Pi_p(i2, sm);
    double b2 = P.Pi_p(i1, sm) * (2.0 * P.Pi_p(i2, sm) + dx1 * P.Pi_pp_3(i2, sm)) - P.Pi(i1, sm) * P.Pi_pp_3(i2, sm);
    b2 = 0.5 * b2 / P.Pi(i2, sm) / P.Pi_p(i2, sm);
    auto bk = [&](int ik, double dxk) {
        const double D = P.Lag(i2, ik, sm) + dxk * P.Lag_p(i2, ik, sm);
        const double b = P.Lag(i1, ik, sm) * (P.Lag(i2, ik, sm) + 2 * dx1 * P.Lag_p(i2, ik, sm)) -
                         dx1 * P.Lag_p(i1, ik, sm) * (P.Lag(i2, ik, sm) + dx1 * P.Lag_p(i2, ik, sm));
        return -b / dxk / D;
    };
    coef[0] = bk(i1, dx1); coef[1] = b2; coef[2] = bk(i3, dx3); coef[3] = bk(i4, dx4); coef[4] = a2;
}

}  // namespace

void int2_build_tables(const DerTables &g, const std::vector<double> &x, int ibc, Int2Tables &out) {
    if (g.ndl != 3 || g.ndr != 5) throw std::runtime_error("direct Poisson: needs a second derivative with 3 LHS and 5 RHS diagonals (CompactDirect6)");
    if (g.periodic) throw std::runtime_error("direct Poisson: the wall-normal direction must not be periodic");
    const int nx = g.n, ndl = 3, ndr = 5, idl = 2, idr = 3;
    if (nx < 10) throw std::runtime_error("direct Poisson: too few points in y");
    if ((int)x.size() != nx) throw std::runtime_error("direct Poisson: the y plan carries no nodes (tlab_fdm_plan_set_aux)");
    if (ibc < 0 || ibc > 3) throw std::runtime_error("direct Poisson: bad boundary type");
    out.n = nx;
    out.bc = ibc;
    std::vector<double> A(g.lhs.begin(), g.lhs.begin() + (size_t)nx * ndl);  // fdmi%rhs (:393)
    double rhsr_b[5 * 8] = {0}, rhsr_t[5 * 8] = {0};
    Reduce red{nx, ndl, ndr, A.data(), g.rhs.data(), rhsr_b, rhsr_t};
    red.run(1);                                                              // FDM_Bcs_Reduce(BCS_BOTH, ...) (:395)
    red.run(2);
#define AA(i, k) A[((i)-1) + (size_t)nx * ((k)-1)]
#define A0(i, k) g.lhs[((i)-1) + (size_t)nx * ((k)-1)]
#define BB(i, k) g.rhs[((i)-1) + (size_t)nx * ((k)-1)]
#define RRB(j, c) rhsr_b[((j)-1) + 5 * (c)]
#define RRT(r, c) rhsr_t[(r) + 5 * ((c)-1)]
    double rhs_b[5 * 8] = {0}, rhs_t[5 * 8] = {0};
#define RB(j, c) rhs_b[((j)-1) + 5 * (c)]
#define RT(r, c) rhs_t[(r) + 5 * ((c)-1)]
    for (int j = 1; j <= idl + 1; ++j)                                        // :400-403
        for (int c = 1; c <= ndl; ++c) RB(j, c) = AA(j, c);
    for (int ir = 1; ir <= idr - 1; ++ir) RB(1 + ir, idl - ir) = -RRB(1 + ir, idr - ir);
    for (int r = 0; r <= idl; ++r)                                            // :405-408
        for (int c = 1; c <= ndl; ++c) RT(r, c) = AA(nx - idl + r, c);
    for (int ir = 1; ir <= idr - 1; ++ir) RT(idl - ir, idl + ir) = -RRT(idr - ir, idr + ir);
    // lhs = B - lambda2 A (:412-432) as operand pairs; the device does the arithmetic per mode
    out.Bt.assign((size_t)nx * 5, 0.0);
    out.A5.assign((size_t)nx * 5, 0.0);
    out.s.assign((size_t)nx, 1.0);
#define BT(i, k) out.Bt[(size_t)((i)-1) * 5 + ((k)-1)]
#define A5(i, k) out.A5[(size_t)((i)-1) * 5 + ((k)-1)]
    for (int i = 1; i <= nx; ++i) {
        for (int k = 1; k <= ndr; ++k) BT(i, k) = BB(i, k);
        A5(i, idr) = A0(i, idl);
        for (int ii = 1; ii <= idl - 1; ++ii) {
            if (i >= 1 + ii) A5(i, idr - ii) = A0(i, idl - ii);
            if (i <= nx - ii) A5(i, idr + ii) = A0(i, idl + ii);
        }
    }
    for (int ir = 1; ir <= idr - 1; ++ir) {                                   // :422-425, :429-432
        for (int c = 1; c <= ndr; ++c) { BT(1 + ir, c) = RRB(1 + ir, c); A5(1 + ir, c) = 0.0; }
        for (int c = 0; c < ndl; ++c) A5(1 + ir, idr - idl + 1 + c) = RB(1 + ir, 1 + c);
        for (int c = 1; c <= ndr; ++c) { BT(nx - ir, c) = RRT(idr - ir, c); A5(nx - ir, c) = 0.0; }
        for (int c = 0; c < ndl; ++c) A5(nx - ir, idr - idl + 1 + c) = RT(idl - ir, 1 + c);
    }
    for (int k = 1; k <= ndr; ++k) { BT(1, k) = 0.0; A5(1, k) = 0.0; BT(nx, k) = 0.0; A5(nx, k) = 0.0; }     // rows 1, n are not part of the system
    // Neumann ends through the 4th-order biased first derivative (:436-514)
    for (int q = 0; q < 3; ++q) out.c1[q] = out.cn[q] = 0.0;
    out.e1 = out.en = 0.0;
    out.nb[0] = out.nb[1] = out.nt[0] = out.nt[1] = 0.0;
    if (ibc == BCS_ND || ibc == BCS_NN) {
        double coef[5];
        coef_c1n4_biased(x, 1, false, coef);
        for (int q = 0; q < 3; ++q) out.c1[q] = -coef[1 + q] / coef[0];        // lhs(1, 1:3)
        for (int c = 0; c <= 7; ++c) RB(1, c) = 0.0;
        RB(1, idl) = 1.0 / coef[0];
        RB(1, idl + 1) = -coef[4] / coef[0];
        out.e1 = RB(1, idl + 1);                                               // lhs(1,1) += lambda2 * rhs_b(1, idl+1)
        for (int ir = 1; ir <= idr - 1; ++ir) {
            out.nb[ir - 1] = RB(1 + ir, idl - ir);                             // lhs(1+ir, idr-ir+1 : idr-ir+3) -= rhs_b(1+ir, idl-ir) * lhs(1, 1:3)
            RB(1 + ir, idl - ir + 1) = RB(1 + ir, idl - ir + 1) + RB(1 + ir, idl - ir) * RB(1, idl + 1);
            RB(1 + ir, idl - ir) = RB(1 + ir, idl - ir) * RB(1, idl);
        }
    }
    if (ibc == BCS_DN || ibc == BCS_NN) {
        double coef[5];
        coef_c1n4_biased(x, nx, true, coef);
        out.cn[0] = -coef[3] / coef[0]; out.cn[1] = -coef[2] / coef[0]; out.cn[2] = -coef[1] / coef[0];      // lhs(nx, ndr-2:ndr)
        for (int c = 1; c <= 8; ++c) RT(idl, c) = 0.0;
        RT(idl, idl) = 1.0 / coef[0];
        RT(idl, idl - 1) = -coef[4] / coef[0];
        out.en = RT(idl, idl - 1);
        for (int ir = 1; ir <= idr - 1; ++ir) {
            out.nt[ir - 1] = RT(idl - ir, idl + ir);                           // lhs(nx-ir, ir : ir+2) -= rhs_t(idl-ir, idl+ir) * lhs(nx, ndr-2:ndr)
            RT(idl - ir, idl + ir - 1) = RT(idl - ir, idl + ir - 1) + RT(idl - ir, idl + ir) * RT(idl, idl - 1);
            RT(idl - ir, idl + ir) = RT(idl - ir, idl + ir) * RT(idl, idl);
        }
    }
    // normalisation (:518-540): rows 2 .. nx-1 only
    const int mx = std::max(idr, idl + 1);
    for (int ir = 2; ir <= mx; ++ir) {
        double dummy = 1.0 / AA(ir, idl);
        for (int c = 0; c <= ndl; ++c) RB(ir, c) = RB(ir, c) * dummy;
        dummy = 1.0 / AA(nx - ir + 1, idl);
        for (int c = 1; c <= ndl + 1; ++c) RT(idl - ir + 1, c) = RT(idl - ir + 1, c) * dummy;
        dummy = 1.0 / AA(ir, idl);
        out.s[ir - 1] = dummy;
        for (int c = 1; c <= ndl; ++c) AA(ir, c) = AA(ir, c) * dummy;
        dummy = 1.0 / AA(nx - ir + 1, idl);
        out.s[nx - ir] = dummy;
        for (int c = 1; c <= ndl; ++c) AA(nx - ir + 1, c) = AA(nx - ir + 1, c) * dummy;
    }
    for (int ir = mx + 1; ir <= nx - mx; ++ir) {
        const double dummy = 1.0 / AA(ir, idl + 1);
        out.s[ir - 1] = dummy;
        for (int c = 1; c <= ndl; ++c) AA(ir, c) = AA(ir, c) * dummy;
    }
    out.R.assign((size_t)nx * 3, 0.0);
    for (int i = 1; i <= nx; ++i)
        for (int k = 1; k <= 3; ++k) out.R[(size_t)(i - 1) * 3 + (k - 1)] = AA(i, k);
    for (int j = 1; j <= 3; ++j)
        for (int c = 0; c <= 3; ++c) out.rb[j - 1][c] = RB(j, c);
    for (int r = 0; r <= 2; ++r)
        for (int c = 1; c <= 4; ++c) out.rt[r][c - 1] = RT(r, c);
#undef BT
#undef A5
#undef AA
#undef A0
#undef BB
#undef RRB
#undef RRT
#undef RB
#undef RT
}

}  // namespace tlab

// include/tlab_amd.h: debug aid (host only)
#include "../../include/tlab_amd.h"
#include "int1_generic.hpp"
#include "plan.hpp"
extern void tlab_set_error(const std::string &s);
extern "C" int tlab_debug_int1_tables(tlab_fdm_plan_t gy, int ibc, int nm, const double *lam, double *fac, double *rb, double *rt, double *R) {
    try {
        if (!gy || !lam || !fac || !rb || !rt || !R || nm < 1 || (ibc != 1 && ibc != 2)) throw std::runtime_error("tlab_debug_int1_tables: bad arguments");
        tlab::Int1Gen G;
        tlab::int1_generic_build(gy->t.der1, ibc, lam, nm, 1.0, G);
        std::copy(G.fac.begin(), G.fac.end(), fac);
        std::copy(G.rb.begin(), G.rb.end(), rb);
        std::copy(G.rt.begin(), G.rt.end(), rt);
        std::copy(G.R.begin(), G.R.end(), R);
        return TLAB_OK;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

// The pack-layout map of the x-transforms (tlab_poisson_fft_x_packed): element (line, kx) of the complex slab at off[kx] + line * width[kx] complex
// values of the pack buffer, i.e. what tlab_pencil_repack_blocks writes for the same block map.  Host arithmetic only.
extern "C" int tlab_debug_pack_map(int nxh, int nblocks, const int *start, const long long *base, long long *off, int *width) {
    if (nxh < 1 || nblocks < 1 || nblocks > 16 || !start || !base || !off || !width || start[0] != 0) { tlab_set_error("tlab_debug_pack_map: bad arguments"); return TLAB_EINVAL; }
    for (int b = 0; b + 1 < nblocks; ++b)
        if (start[b + 1] < start[b] || start[b + 1] > nxh) { tlab_set_error("tlab_debug_pack_map: block starts must increase within [0, nx/2+1]"); return TLAB_EINVAL; }
    for (int b = 0; b < nblocks; ++b) {
        const int e = b + 1 < nblocks ? start[b + 1] : nxh;
        for (int i = start[b]; i < e; ++i) { off[i] = base[b] + (i - start[b]); width[i] = e - start[b]; }
    }
    return TLAB_OK;
}
