// Lambda-independent tables of the first-order integral operators u' + lambda u = f (see poisson_host.cpp).
#pragma once
#include <vector>

#include "fdm_schemes.hpp"

namespace tlab {

struct Int1Tables {
    int n = 0;
    int bc = 0;                  // 1 = BCS_MIN (value given at the bottom), 2 = BCS_MAX
    std::vector<double> L0, L1;  // [n][5] row-major: lhs(lambda) = L0 + lambda L1, before the opposite-end reduction
    std::vector<double> R;       // [n][3] row-major: rhs (A, reduced + normalised)
    double rb[3][4];             // rhs_b(1:3, 0:3): final for BCS_MIN; for BCS_MAX rebuilt per mode from R
    double rt[3][4];             // rhs_t(0:2, 1:4): final for BCS_MAX; for BCS_MIN rebuilt per mode from R
};

// ibc: 1 = BCS_MIN, 2 = BCS_MAX.  g: first-derivative tables of the (non-periodic) y direction.
void int1_build_tables(const DerTables &g, int ibc, Int1Tables &out);

}  // namespace tlab
