#!/usr/bin/env python3
"""Fixtures for the direct FIRST-derivative schemes (SpaceOrder1 = CompactDirect4 / CompactDirect6: FDM_C1N4_Direct, FDM_C1N6_Direct of
fdm_comx_direct.f90 with g%matmul => MatMul_3d / MatMul_5d, fdm_derivative.f90:123-129), made by the reference itself (oracle/_ref):
the y-plan tables FDM_CreatePlan builds on a tanh-stretched grid with (mode1, mode2) = (17, 17) and (16, 16) -- the coefficient formulas are
NOT restated anywhere in this repository -- and the outputs of OPR_Partial_Y (all types, all four boundary variants) and OPR_Burgers_Y.

    make -C oracle && python3 tests/golden/make_golden_direct1.py"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

KEYS = ("ndl1", "ndr1", "ndl2", "ndr2", "need_1der", "lhs1", "rhs1", "lu1", "rhs_b1", "rhs_t1", "mwn1", "lhs2", "rhs2", "lu2", "mwn2", "jac")


def ygrid(ny):
    return 0.5 * (1 + np.tanh(2 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(2))


if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    out = {}
    nx, nz = 8, 8
    x, z = np.arange(nx) / nx, np.arange(nz) / nz * 2.0
    rng = np.random.default_rng(20250611)
    for ny in (24, 72, 128):          # 128: tables only (register-tile / half-wave-tile kernels against the oracle)
        if ny != 128:
            out["ny%d_u" % ny] = rng.uniform(-1, 1, nx * ny * nz)
            out["ny%d_v" % ny] = rng.uniform(-1, 1, nx * ny * nz)
        out["ny%d_nodes" % ny] = ygrid(ny)
    # Order matters.  MatMul_3d / MatMul_5d read one more row of g%rhs_b / g%rhs_t (rhs_b(idr+1,:), rhs_t(0,:)) than FDM_Bcs_Neumann sets for
    # a first derivative (it fills idr rows, fdm_base.f90:224, :262): in a program that creates its plan once those rows hold the initial
    # zeros of the plan object, so that row idr+1 from a Neumann wall gets a zero right-hand side (reference defect).  The
    # library behind oracle/_ref keeps ONE plan object per direction for the life of the process, so the 3-diagonal scheme goes first:
    # after a 5-diagonal one its never-set row 3 would hold the stale row of that scheme instead of the zeros of a fresh plan.
    for mode in (17, 16):
        for ny in (24, 72, 128):
            u, v, y = out.get("ny%d_u" % ny), out.get("ny%d_v" % ny), out["ny%d_nodes" % ny]
            R.init(nx, ny, nz)
            R.fdm_create(1, x, True, True, mode, mode)
            R.fdm_create(2, y, False, False, mode, mode)
            R.fdm_create(3, z, True, True, mode, mode)
            tab = R.fdm_arrays(2, ny)
            pre = "m%d_ny%d_" % (mode, ny)
            for k in KEYS:
                out[pre + k] = np.asarray(tab[k])
            assert not tab["rhs_b1"][3 if mode == 16 else 2].any() and not tab["rhs_t1"][0].any(), "stale rows in the plan object"
            for ibc in ((0, 1, 2, 3) if ny != 128 else ()):
                for t in ((1, 2, 3) if (ny == 24 and ibc == 0) else (1, 3) if ny == 24 else (1,)):      # (OPR_P2 alone does not see ibc)
                    r, t1 = R.partial(2, t, nx, ny, nz, ibc, u)
                    out[pre + "partial_t%d_bc%d" % (t, ibc)] = r
                    if t == 3:
                        out[pre + "partial_t3_bc%d_tmp1" % ibc] = t1
                if ny == 24 or ibc == 0:
                    out[pre + "burgers_bc%d" % ibc], _ = R.burgers(2, nx, ny, nz, ibc, 1.0 / 300.0, u, v)
    out["visc"] = 1.0 / 300.0
    out["nx"], out["nz"] = nx, nz
    np.savez_compressed(os.path.join(HERE, "direct1_y.npz"), **out)
    print("wrote direct1_y.npz", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KiB raw")
