#!/usr/bin/env python3
"""Fixture for [Staggering] StaggerHorizontalPressure, made by the reference itself (oracle/_ref): the interpolation tables FDM_CreatePlan
adds to a periodic direction (g%intl%lu0i, lu1i, fdm.f90:236-248 -> FDM_Interpol_Initialize, fdm_interpol.f90), the interpolatory der1%mwn,
and OPR_Partial_X / OPR_Partial_Z with the types OPR_P1_INT_VP, OPR_P1_INT_PV, OPR_P0_INT_VP, OPR_P0_INT_PV (opr_partial.f90:213-227) on a
random field.

    make -C oracle && python3 tests/golden/make_golden_stagger.py"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    nx, ny, nz = 32, 6, 24
    rng = np.random.default_rng(20250703)
    x = np.arange(nx) / nx * 2.0 * np.pi
    z = np.arange(nz) / nz * 3.0
    y = np.arange(ny) / (ny - 1.0)
    R.init(nx, ny, nz)
    R.set_stagger(True)
    R.fdm_create(1, x, True, True)
    R.fdm_create(2, y, False, True)
    R.fdm_create(3, z, True, True)
    out = {"x": x, "y": y, "z": z}
    u = rng.uniform(-1, 1, (nz, ny, nx))
    out["u"] = u
    for d, n in ((1, nx), (3, nz)):
        lu0i, lu1i = R.intl_arrays(d, n)
        out["plan%d_lu0i" % d], out["plan%d_lu1i" % d] = lu0i, lu1i
        out["plan%d_mwn1" % d] = R.fdm_arrays(d, n)["mwn1"]
        for t in (5, 6, 7, 8):
            out["d%d_t%d" % (d, t)] = R.partial(d, t, nx, ny, nz, 0, u)[0]
    R.set_stagger(False)
    np.savez_compressed(os.path.join(HERE, "stagger.npz"), **out)
    print("wrote stagger.npz")
