#!/usr/bin/env python3
"""Fixtures for the direct second-derivative scheme (SpaceOrder2 = CompactDirect6, fdm_comx_direct.f90; examples/Case81-93), made by the
reference itself (oracle/_ref): the y-plan tables FDM_CreatePlan builds for a tanh-stretched grid with (mode1, mode2) = (6, 16) at several
sizes -- the coefficient formulas are NOT restated anywhere in this repository -- and, for the smallest size, the operator outputs.

    make -C oracle && python3 tests/golden/make_golden_direct.py"""
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
    nx, nz = 16, 8
    x, z = np.arange(nx) / nx, np.arange(nz) / nz * 2.0
    for ny in (24, 64, 128, 512):
        R.init(nx, ny, nz)
        y = ygrid(ny)
        R.fdm_create(1, x, True, True, 6, 16)
        R.fdm_create(2, y, False, False, 6, 16)
        R.fdm_create(3, z, True, True, 6, 16)
        tab = R.fdm_arrays(2, ny)
        for k in KEYS:
            out["ny%d_%s" % (ny, k)] = np.asarray(tab[k])
        out["ny%d_nodes" % ny] = y
        if ny == 24:
            rng = np.random.default_rng(20250509)
            u = rng.uniform(-1, 1, nx * ny * nz); v = rng.uniform(-1, 1, nx * ny * nz)
            out["u"], out["v"], out["visc"] = u, v, 1.0 / 300.0
            for t in (1, 2, 3):
                r, t1 = R.partial(2, t, nx, ny, nz, 0, u)
                out["partial_t%d" % t] = r
                if t == 3:
                    out["partial_t3_tmp1"] = t1
            out["burgers"], _ = R.burgers(2, nx, ny, nz, 0, out["visc"], u, v)
            tx = R.fdm_arrays(1, nx)
            out["x_ndr2"] = tx["ndr2"]           # periodic directions fall back to the Jacobian-hyper scheme (fdm.f90:155-158): 7
    np.savez_compressed(os.path.join(HERE, "direct_y.npz"), **out)
    print("wrote direct_y.npz", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KiB raw")
