#!/usr/bin/env python3
"""Golden vectors for the per-Fourier-mode arithmetic of OPR_Poisson (FDM_Int1_* and OPR_ODE2_Factorize_NN/_NN_Sing),
generated from the reference's own Fortran through oracle/_ref (see make_golden.py).  The top-level OPR_Poisson
cannot be built in this image (needs fftw3.f03), so there is no reference-generated full-field Poisson fixture."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402


def case(name, n, stretch, mode1=6):
    """mode1: SpaceOrder1 of the y plan the integral operators invert (fdm_derivative.f90:52-58): 6 CompactJacobian6 (3 / 5 diagonals: pentadiagonal
    integral systems, PENTADFS), 5 CompactJacobian6Penta (5 / 7: HEPTADFS), 4 CompactJacobian4 (3 / 3: TRIDFS) -- fdm_integral.f90:75-83."""
    y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2)) if stretch else np.arange(n) / (n - 1.0) * 2.0
    R.init(4, n, 4)
    R.fdm_create(2, y, False, not stretch, mode1=mode1)
    ndl, ndr = {6: (3, 5), 5: (5, 7), 4: (3, 3)}[mode1]
    rng = np.random.default_rng(20250509 + n)
    lams = np.array([0.0, 1.2246467991473532e-16, 0.5, 6.283185307179586, 97.0, 1500.0])
    out = {"y": y, "uniform": int(not stretch), "lams": lams, "mode1": mode1}
    for il, lam in enumerate(lams):
        for ibc, sgn in ((1, 1.0), (2, -1.0)):
            R.int1_create(sgn * lam, ibc, False)
            for k, v in R.int1_tables(n, ibc, ndr, ndl).items():
                out["sys_l%d_bc%d_%s" % (il, ibc, k)] = v
            R.int1_create(sgn * lam, ibc, True)
            out["lu_l%d_bc%d_lhs" % (il, ibc)] = R.int1_tables(n, ibc, ndr, ndl)["lhs"]
            f = rng.uniform(-1, 1, (n, 3))
            res = np.zeros((n, 3))
            res[0 if ibc == 1 else n - 1] = rng.uniform(-1, 1, 3)
            r, du = R.int1_solve(ibc, f, res)
            out["int1_l%d_bc%d_f" % (il, ibc)] = f
            out["int1_l%d_bc%d_res0" % (il, ibc)] = res
            out["int1_l%d_bc%d_res" % (il, ibc)] = r
            out["int1_l%d_bc%d_du" % (il, ibc)] = du
        f = rng.uniform(-1, 1, (n, 2))
        bcs = rng.uniform(-1, 1, (2, 2))
        itype = 1 if lam > 1e-10 else 2
        u, v = R.ode2(itype, lam, f, bcs)
        out["ode2_l%d_type" % il] = itype
        out["ode2_l%d_f" % il] = f
        out["ode2_l%d_bcs" % il] = bcs
        out["ode2_l%d_u" % il] = u
        out["ode2_l%d_v" % il] = v
    # OPR_ODE2_Factorize_DD / _DD_Sing (ibc = BCS_DD of OPR_Poisson, opr_elliptic.f90:322-329); own generator: the entries above stay as they were
    rng = np.random.default_rng(20250511 + n)
    for il, lam in enumerate(lams):
        f = rng.uniform(-1, 1, (n, 2))
        bcs = rng.uniform(-1, 1, (2, 2))
        itype = 3 if lam > 1e-10 else 4
        u, v = R.ode2(itype, lam, f, bcs)
        out["ode2dd_l%d_type" % il] = itype
        out["ode2dd_l%d_f" % il] = f
        out["ode2dd_l%d_bcs" % il] = bcs
        out["ode2dd_l%d_u" % il] = u
        out["ode2dd_l%d_v" % il] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def main():
    import sys as _sys
    if "--all" in _sys.argv:              # the two fixtures of round 1 (regenerating them reproduces the committed files bit for bit)
        case("poisson_modes_stretched_40", 40, True)
        case("poisson_modes_uniform_32", 32, False)
    case("poisson_modes_penta_stretched_40", 40, True, mode1=5)          # 7-diagonal integral systems (HEPTADFS / HEPTADSS, MatMul_5d)
    case("poisson_modes_jacobian4_stretched_36", 36, True, mode1=4)      # 3-diagonal ones (TRIDFS / TRIDSS, MatMul_3d)


if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    main()
