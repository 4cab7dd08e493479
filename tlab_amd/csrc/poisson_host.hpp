// Lambda-independent tables of the first-order integral operators u' + lambda u = f (see poisson_host.cpp).
#pragma once
#include <vector>

#include "fdm_schemes.hpp"

namespace tlab {

struct Int1Tables {
    int n = 0;
    int bc = 0;                  // 1 = BCS_MIN (value given at the bottom), 2 = BCS_MAX
    std::vector<double> L0, L1;  // [n][5] row-major: lhs(lambda) = (L0 + lambda L1) * S before the opposite-end reduction; S [n] (the row
                                 // normalisation of fdm_integral.f90:175-201) is stored behind L0's [n][5] block
    std::vector<double> R;       // [n][3] row-major: rhs (A, reduced + normalised)
    double rb[3][4];             // rhs_b(1:3, 0:3): final for BCS_MIN; for BCS_MAX rebuilt per mode from R
    double rt[3][4];             // rhs_t(0:2, 1:4): final for BCS_MAX; for BCS_MIN rebuilt per mode from R
};

// ibc: 1 = BCS_MIN, 2 = BCS_MAX.  g: first-derivative tables of the (non-periodic) y direction.
void int1_build_tables(const DerTables &g, int ibc, Int1Tables &out);

// Second-order integral operator of the DIRECT elliptic solver, (B - lambda2 A) u = A f (FDM_Int2_CreateSystem, fdm/fdm_integral.f90:366-557).
// The solution of the Neumann problem is sensitive to the LAST BIT of the matrix entries (the row sums carry lambda2 h^2 ~ 1e-4 against O(1)
// entries: an affine table L0 + lambda2 L1 with the normalisation folded in reproduces the reference to 5e-12 only, measured), so the tables
// keep the reference's operands and the device repeats its operations in its order, without fused multiply-adds:
//     row j of lhs(lambda2) = (Bt[j][:] - lambda2 * A5[j][:]) [- Neumann term, rows 2, 3, n-2, n-1] * s[j]
// rhs, rhs_b, rhs_t do not depend on lambda2.
struct Int2Tables {
    int n = 0;
    int bc = 0;                  // BCS_DD / BCS_ND / BCS_DN / BCS_NN = 0..3
    std::vector<double> Bt, A5;  // [n][5] row-major: B (rows 2, 3, n-2, n-1: its boundary-reduced rows) and the lambda2 operand of every row
    std::vector<double> s;       // [n] row normalisation (1 at rows 1, n)
    std::vector<double> R;       // [n][3] row-major: fdmi%rhs (A, reduced at both ends + normalised)
    double rb[3][4];             // rhs_b(1:3, 0:3)
    double rt[3][4];             // rhs_t(0:2, 1:4)
    // Neumann ends (:436-514): row 1 = (c1[0] + lambda2 e1, c1[1], c1[2]) on p_2..p_4, rows 2, 3 get - nb[ir-1] * row 1; same at the top
    double c1[3], e1, nb[2];
    double cn[3], en, nt[2];     // row n = (cn[0], cn[1], cn[2] + lambda2 en) on p_{n-3}..p_{n-1}; rows n-1, n-2 get - nt[ir-1] * row n
};

// g: second-derivative tables of the (non-periodic) y direction with a tridiagonal A and a pentadiagonal B (CompactDirect6); x: its nodes
void int2_build_tables(const DerTables &g, const std::vector<double> &x, int ibc, Int2Tables &out);

}  // namespace tlab
