#!/usr/bin/env python3
"""Fixtures for the 1-D filters behind [Dealiasing] (OPR_FILTER_1D, operators/opr_filter.f90:393-460), made by the reference's own filter
modules (src/filters/flt_compact.f90, flt_explitic.f90 through oracle/_ref, oracle/ref_driver_filter.f90): the coefficient tables
OPR_FILTER_INITIALIZE builds (f%coeffs; their generators are not restated in this repository) and filtered random lines, for the types
compact (1), explicit6 (2), explicit4 (3), compactcutoff (9), periodic and non-periodic with several end conditions.

    make -C oracle && python3 tests/golden/make_golden_filters.py"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402
from oracle import tlab_oracle as O  # noqa: E402

if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    out = {}
    rng = np.random.default_rng(20250620)
    R.init(64, 64, 8)
    cases = []
    for n in (24, 64):
        y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(1.5))
        x = np.arange(n) / n * 2.0
        jy, jx = O.FdmPlan(y, False, False).jac[:, 0], O.FdmPlan(x, True, True).jac[:, 0]
        out["n%d_y" % n], out["n%d_x" % n], out["n%d_jy" % n], out["n%d_jx" % n] = y, x, jy, jx
        u = rng.uniform(-1, 1, (n, 6))
        out["n%d_u" % n] = u
        for t in (1, 2, 3, 9):
            for per, nodes, jac, bcs in ((True, x, jx, [(0, 0)]), (False, y, jy, [(1, 1), (2, 6), (6, 2), (6, 6)])):
                for b0, b1 in bcs:
                    key = "n%d_t%d_p%d_b%d%d" % (n, t, int(per), b0, b1)
                    c = R.filter_init(t, nodes, jac, per, b0, b1)
                    out[key + "_coeffs"] = c
                    out[key + "_res"] = R.filter_1d(t, per, b0, b1, c, u)
                    cases.append(key)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "filters.npz"), **out)
    print("wrote filters.npz", len(cases), "cases")
