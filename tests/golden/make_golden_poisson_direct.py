#!/usr/bin/env python3
"""Golden vectors for the per-Fourier-mode arithmetic of the DIRECT elliptic solver (EllipticOrder = CompactDirect6,
OPR_Poisson_FourierXZ_Direct): FDM_Int2_Initialize / FDM_Int2_Solve (fdm/fdm_integral.f90:334,626) run by the reference's own Fortran
through oracle/_ref on the y plan FDM_CreatePlan builds with (mode1, mode2) = (CompactJacobian6, CompactDirect6), for all four boundary
types and a range of lambda.  The plan tables travel with the fixture (the direct-scheme formulas are not restated in this repository).

    make -C oracle && python3 tests/golden/make_golden_poisson_direct.py"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

KEYS = ("ndl1", "ndr1", "ndl2", "ndr2", "need_1der", "lhs1", "rhs1", "lu1", "rhs_b1", "rhs_t1", "mwn1", "lhs2", "rhs2", "lu2", "mwn2", "jac")


def case(name, n, mode2=16):
    y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2))
    R.init(4, n, 4)
    R.fdm_create(2, y, False, False, 6, mode2)
    tab = R.fdm_arrays(2, n)
    out = {"tab_" + k: np.asarray(tab[k]) for k in KEYS}
    out["y"] = y
    out["mode2"] = mode2
    rng = np.random.default_rng(20250510 + n)
    lams = np.array([0.0, 1e-3, 0.5, 39.47841760435743, 3.0e3, 2.6e5])
    out["lams"] = lams
    for il, lam in enumerate(lams):
        for ibc in (0, 1, 2, 3):                 # BCS_DD, BCS_ND, BCS_DN, BCS_NN
            if lam == 0.0 and ibc == 3:
                continue                          # singular: the reference never solves it (opr_elliptic.f90:236-240 uses BCS_DN)
            R.int2_create(lam, ibc)
            for k, v in R.int2_tables(n).items():
                out["lu_l%d_bc%d_%s" % (il, ibc, k)] = v
            f = rng.uniform(-1, 1, (n, 2))
            res = np.zeros((n, 2))
            res[0], res[n - 1] = rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 2)
            out["int2_l%d_bc%d_f" % (il, ibc)] = f
            out["int2_l%d_bc%d_res0" % (il, ibc)] = res
            out["int2_l%d_bc%d_res" % (il, ibc)] = R.int2_solve(f, res)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    case("poisson_direct_modes_24", 24)
    case("poisson_direct_modes_96", 96)
    case("poisson_direct_modes_c4_40", 40, mode2=17)      # EllipticOrder = CompactDirect4 (FDM_C2N4_Direct: same (3, 5) diagonals, opr_elliptic.f90:113-116)
