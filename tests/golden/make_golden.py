#!/usr/bin/env python3
"""Generate golden vectors from the reference itself (oracle/_ref/libtlab_ref.so = the reference's Fortran
sources compiled in place by oracle/Makefile).  Run in the build container, where /root/reference exists:

    make -C oracle && python3 tests/golden/make_golden.py

Writes tests/golden/*.npz (inputs + plan arrays + expected outputs; data only, no reference source).
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

SEED = 20250509


def grids(nx, ny, nz, ystretch):
    x = np.arange(nx) / nx
    z = np.arange(nz) / nz * 2.0
    if ystretch:
        y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(2))
    else:
        y = np.arange(ny) / (ny - 1) * 1.5
    return x, y, z


def derivs_case(name, nx, ny, nz, ystretch, mode1=6, mode2=7):
    R.init(nx, ny, nz)
    x, y, z = grids(nx, ny, nz, ystretch)
    spec = {1: (x, True, True), 2: (y, False, not ystretch), 3: (z, True, True)}
    out = {"nx": nx, "ny": ny, "nz": nz, "mode1": mode1, "mode2": mode2, "x": x, "y": y, "z": z,
           "yuniform": int(not ystretch)}
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
    out["u"], out["v"] = u, v
    visc = 1.0 / 500.0
    out["visc"] = visc
    for d in (1, 2, 3):
        for ibc in ((0, 1, 2, 3) if d == 2 else (0,)):
            for t in (1, 2, 3):
                r, t1 = R.partial(d, t, nx, ny, nz, ibc, u)
                out["partial_d%d_t%d_bc%d" % (d, t, ibc)] = r
                if t == 3:
                    out["partial_d%d_t%d_bc%d_tmp1" % (d, t, ibc)] = t1
            r, t1 = R.burgers(d, nx, ny, nz, ibc, visc, u, v)
            out["burgers_d%d_bc%d" % (d, ibc)] = r
            if d != 3:
                out["burgers_d%d_bc%d_tmp1" % (d, ibc)] = t1
    for ibc in (1, 2, 3):               # BOUNDARY_BCS_NEUMANN_Y wall planes (ref_driver.f90: ref_bcs_neumann_y)
        hb, ht = R.bcs_neumann_y(ibc, nx, ny, nz, u)
        out["bcsn_bc%d_hb" % ibc], out["bcsn_bc%d_ht" % ibc] = hb, ht
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KiB raw")


if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing: run `make -C oracle` where /root/reference exists")
    derivs_case("derivs_stretched_16x12x8", 16, 12, 8, True)
    derivs_case("derivs_uniform_24x14x10", 24, 14, 10, False)
    derivs_case("derivs_c2n6_16x12x8", 16, 12, 8, True, 6, 6)
    try:
        import make_golden_poisson  # noqa: F401  (added with the Poisson milestone)
        make_golden_poisson.main()
    except ImportError:
        pass
