#!/usr/bin/env python3
"""Golden vectors of SpaceOrder1 = CompactJacobian6Penta (fdm_com1_jacobian.f90:136-192; pentadiagonal LHS, 7-diagonal RHS; PENTADFS2 /
PENTADSS2 / PENTADPFS / PENTADPSS, utils/linear5.f90:156-411; MatMul_7d_antisym, fdm_matmul.f90:491-558) from the reference itself
(oracle/_ref).  Run in the build container:  python3 tests/golden/make_golden_penta.py  ->  tests/golden/derivs_penta_16x12x8.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from oracle import ref_lib as R  # noqa: E402
from make_golden import grids, SEED  # noqa: E402

if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    nx, ny, nz, mode1, mode2 = 16, 12, 8, 5, 6          # FDM_COM6_JACOBIAN_PENTA, second derivative FDM_COM6_JACOBIAN
    R.init(nx, ny, nz)
    x, y, z = grids(nx, ny, nz, True)
    spec = {1: (x, True, True), 2: (y, False, False), 3: (z, True, True)}
    out = {"nx": nx, "ny": ny, "nz": nz, "mode1": mode1, "mode2": mode2, "x": x, "y": y, "z": z, "yuniform": 0}
    for d, (nodes, per, uni) in spec.items():
        R.fdm_create(d, nodes, per, uni, mode1, mode2)
        for k, v in R.fdm_arrays(d, len(nodes)).items():
            out["plan%d_%s" % (d, k)] = v
    rng = np.random.default_rng(SEED)
    N = nx * ny * nz
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    smooth = (np.sin(2 * np.pi * X) * np.cos(4 * np.pi * Y) * np.sin(np.pi * Z)).transpose(2, 1, 0).ravel()
    u = smooth + 0.1 * rng.uniform(-1, 1, N)
    v = np.roll(smooth, 7) + 0.1 * rng.uniform(-1, 1, N)
    out["u"], out["v"], out["visc"] = u, v, 1.0 / 500.0
    for d in (1, 2, 3):
        for ibc in ((0, 1, 2, 3) if d == 2 else (0,)):
            for t in (1, 2, 3):
                r, t1 = R.partial(d, t, nx, ny, nz, ibc, u)
                out["partial_d%d_t%d_bc%d" % (d, t, ibc)] = r
                if t == 3:
                    out["partial_d%d_t%d_bc%d_tmp1" % (d, t, ibc)] = t1
            r, t1 = R.burgers(d, nx, ny, nz, ibc, out["visc"], u, v)
            out["burgers_d%d_bc%d" % (d, ibc)] = r
            if d != 3:
                out["burgers_d%d_bc%d_tmp1" % (d, ibc)] = t1
    np.savez_compressed(os.path.join(HERE, "derivs_penta_16x12x8.npz"), **out)
    print("wrote derivs_penta_16x12x8.npz")
