// Host-factorized first-order integral operators with 3 or 7 diagonals (see int1_generic.cpp).
#pragma once
#include <vector>

#include "fdm_schemes.hpp"

namespace tlab {

struct Int1Gen {
    int n = 0, ndi = 0, nri = 0, bc = 0;      // rows; diagonals of the integral system (3 | 7) and of its right-hand side (3 | 5); 1 BCS_MIN, 2 BCS_MAX
    long long nm = 0;                         // modes
    std::vector<double> fac;                  // [ndi][n][nm]: fdmi%lhs after FDM_Int1_Initialize -- rows 2..n-1 the LU factors as TRIDFS / HEPTADFS leave
                                              // them, rows 1 and n the reduced boundary rows the solve reads for the free end and the derivative
    std::vector<double> rb, rt;               // [40][nm]: fdmi%rhs_b(1:5, 0:7) [(j-1) + 5 c], fdmi%rhs_t(0:4, 1:8) [r + 5 (c-1)]
    std::vector<double> R;                    // [n][nri] row-major: fdmi%rhs (the same for every mode)
};

bool int1_generic_applies(const DerTables &g);
// lam: nm constants (HOST); the system of mode t is built for lam_sign * lam[t] (opr_elliptic.f90:205-209: +lambda for BCS_MIN, -lambda for BCS_MAX)
void int1_generic_build(const DerTables &g, int ibc, const double *lam, long long nm, double lam_sign, Int1Gen &out);

}  // namespace tlab
