#!/usr/bin/env python3
"""Golden fixture of the Fortran mini-driver test (tlab_amd/fortran/test_rk_driver.f90): two full Runge-Kutta steps (2 x 3 substeps, the
tendency scaling in between) of the incompressible box by the numpy oracle (oracle/tlab_oracle_rhs.py, itself pinned against oracle/_ref),
for a no-slip / Dirichlet case and a free-slip / Neumann case.  Inputs and expected outputs -> tests/golden/rk_step_<case>.npz.

    python tests/golden/make_golden_rk.py
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle.tlab_oracle_rhs import DnsOracle  # noqa: E402

KDT = [1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0]
KCO = [-5.0 / 9.0, -153.0 / 128.0]


def case_setup(name):
    nx, ny, nz = 24, 20, 12
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    rng = np.random.default_rng(2025)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * Y)
    if name == "noslip":
        q = [((np.sin(np.pi * X + k) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(3)]
        bcs = dict(flow_jmin=[3, 3, 3], flow_jmax=[3, 3, 3], scal_jmin=[3], scal_jmax=[3],
                   ini={"VelocityJmin": "noslip", "VelocityJmax": "noslip", "Scalar1Jmin": "dirichlet", "Scalar1Jmax": "dirichlet"})
    else:
        q = [(np.sin(np.pi * X) * np.cos(2 * np.pi * Z) * np.cos(np.pi * Y) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel(),
             ((np.cos(np.pi * X) * np.sin(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel(),
             (np.sin(2 * np.pi * X + 1) * np.sin(2 * np.pi * Z) * np.cos(np.pi * Y) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()]
        bcs = dict(flow_jmin=[4, 3, 4], flow_jmax=[3, 3, 3], scal_jmin=[4], scal_jmax=[3],
                   ini={"VelocityJmin": "freeslip", "VelocityJmax": "noslip", "Scalar1Jmin": "neumann", "Scalar1Jmax": "dirichlet"})
    s = [(np.cos(np.pi * X) * np.cos(np.pi * Y) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()]
    return dict(nx=nx, ny=ny, nz=nz, x=x, y=y, z=z, q=q, s=s, reynolds=800.0, schmidt=0.7, dtime=2e-3, steps=2, **bcs)


def run_oracle(c):
    o = DnsOracle(c["x"], c["y"], c["z"], nscal=1, visc=1.0 / c["reynolds"], schmidt=(c["schmidt"],), yuniform=False)
    o.flow_jmin, o.flow_jmax, o.scal_jmin, o.scal_jmax = c["flow_jmin"], c["flow_jmax"], c["scal_jmin"], c["scal_jmax"]
    o.q = [a.copy() for a in c["q"]]
    o.s = [a.copy() for a in c["s"]]
    for _ in range(c["steps"]):                       # TIME_RUNGEKUTTA, tools/dns/time.f90:212-298
        for a in o.hq + o.hs:
            a[:] = 0.0
        for k in range(3):
            last = k == 2
            o.time_substep(c["dtime"] * KDT[k], 1.0 if last else KCO[k], not last)
    return o.q, o.s


if __name__ == "__main__":
    for name in ("noslip", "freeslip"):
        c = case_setup(name)
        q1, s1 = run_oracle(c)
        out = os.path.join(ROOT, "tests", "golden", "rk_step_%s.npz" % name)
        np.savez_compressed(out, x=c["x"], y=c["y"], z=c["z"], q0=np.array(c["q"]), s0=np.array(c["s"]), q1=np.array(q1), s1=np.array(s1),
                            reynolds=c["reynolds"], schmidt=c["schmidt"], dtime=c["dtime"], steps=c["steps"],
                            ini=np.array(["%s=%s" % kv for kv in c["ini"].items()]))
        print(out, os.path.getsize(out))
