#!/usr/bin/env python3
"""By how much do two legitimate builds of the REFERENCE differ?  

The reference's own Fortran is compiled twice by the committed recipe (oracle/Makefile):
    make -C oracle                 -> oracle/_ref      amdflang -O2, x86-64 baseline: no fused multiply-add exists, every a*b+c rounds twice
    make -C oracle fma             -> oracle/_ref_fma  amdflang -O2 -march=haswell: flang contracts a*b+c into vfmadd (one rounding)
and the same inputs go through both (one subprocess per library: the reference keeps its plans in module variables):

  derivatives   FDM_Der1_Solve / FDM_Der2_Solve (fdm/fdm_derivative.f90:218-278, :413-459) on 512 / 1024 / 2048-point lines, periodic uniform,
                wall-bounded uniform and wall-bounded tanh-stretched (the y grid of BASELINE configs[4])
  poisson       the per-mode stage of OPR_Poisson_FourierXZ_Factorize (operators/opr_elliptic.f90:308-333): FDM_Int1_Initialize +
                OPR_ODE2_Factorize_NN / _NN_Sing (operators/opr_odes.f90:265-386, :165-183; fdm/fdm_integral.f90:58-314) for EVERY Fourier mode of the
                projection forcing of tests/test_gpu_poisson.py::test_projection_forcing_within_the_oracles_own_scatter (64 x 512 x 16, first substep of
                a non-solenoidal field) and of the same forcing from an already projected field (substep 4); transforms by numpy.fft on both sides

What is stored (tests/golden/ref_fma_scatter.npz): per case the relative difference max|a - b| / max|b| between the two builds -- the
reference-derived yardstick the device errors are asserted against NEXT to the oracle's one-ulp scatter (tests/scatter.py::ref_build_bound) -- the
worst modes, and the compiler / flags that made the two libraries.  Nothing here is a device result.

    python tests/golden/make_golden_fma_scatter.py            (needs /root/reference; builds both libraries if missing)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIBS = {"plain": os.path.join(ROOT, "oracle", "_ref", "libtlab_ref.so"), "fma": os.path.join(ROOT, "oracle", "_ref_fma", "libtlab_ref.so")}
LINE_LENGTHS = (512, 1024, 2048)
POISSON_GRID = (64, 512, 16)


def nodes(kind, n):
    if kind == "periodic":
        return np.arange(n) / n, True, True
    if kind == "uniform":
        return np.arange(n) / (n - 1.0), False, True
    return 0.5 * (1.0 + np.tanh(2.0 * (2.0 * np.arange(n) / (n - 1.0) - 1.0)) / np.tanh(2.0)), False, False      # SURVEY 8d, configs[4]


def line_input(n, nlines=16, seed=20250509):
    rng = np.random.default_rng(seed + n)
    t = np.arange(n)[:, None] / n
    k = 1.0 + np.arange(nlines)[None, :]
    return np.sin(2 * np.pi * k * t) * np.cos(4 * np.pi * t) + 0.1 * rng.uniform(-1, 1, (n, nlines))


def projection_forcings():
    """(f, hb, ht) handed to OPR_Poisson by the numpy oracle's RHS: first substep of the non-solenoidal test field, and the first substep of the SECOND
    step (substep 4), when the velocity has been projected three times."""
    import test_gpu_rhs as M
    import oracle.tlab_oracle_rhs as R
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = POISSON_GRID
    x, y, z = M.grids(nx, ny, nz, False)
    q0, s0 = M.init_fields(nx, ny, nz, x, y, z, 23, noise=1e-3)
    o = DnsOracle(x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(0.7,), yuniform=True)
    for i in range(3):
        o.q[i] = q0[i].copy()
    o.s[0] = s0[0].copy()
    caps = []
    orig = R.OP.opr_poisson_fxz

    def spy(plan, f, hb, ht, *a, **k):
        caps.append((f.copy(), hb.copy(), ht.copy()))
        return orig(plan, f, hb, ht, *a, **k)
    R.OP.opr_poisson_fxz = spy
    kdt, kco, dtime = [1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0], [-5.0 / 9.0, -153.0 / 128.0], 1e-3
    try:
        for k in range(4):
            s = k % 3
            if s == 0:
                o.hq = [np.zeros_like(a) for a in o.hq]
                o.hs = [np.zeros_like(a) for a in o.hs]
            o.time_substep(dtime * kdt[s], 1.0 if s == 2 else kco[s], s != 2)
    finally:
        R.OP.opr_poisson_fxz = orig
    return {"first": caps[0], "projected": caps[3]}, o.poisson, y


def worker(out):
    """Runs in a subprocess with TLAB_REF_LIB pointing at one of the two libraries."""
    from oracle import ref_lib as R
    res = {}
    for n in LINE_LENGTHS:
        u = line_input(n)
        for kind in ("periodic", "uniform", "stretched"):
            xs, per, uni = nodes(kind, n)
            R.init(4, n, 4)
            R.fdm_create(2, xs, per, uni)
            d1 = R.der1_solve(2, 0, u)
            d2 = R.der2_solve(2, u, d1)
            res["der1_%s_%d" % (kind, n)] = d1
            res["der2_%s_%d" % (kind, n)] = d2
            res["lu1_%s_%d" % (kind, n)] = R.fdm_arrays(2, n)["lu1"]
    forc, plan, y = projection_forcings()
    nx, ny, nz = POISSON_GRID
    R.init(nx, ny, nz)
    R.fdm_create(2, y, False, True)
    nxh = nx // 2 + 1
    Mm = nz * nxh
    lam = np.sqrt(plan.lam2.reshape(Mm))
    sing = plan.sing.reshape(Mm)
    for tag, (f0, hb, ht) in forc.items():
        a = np.array(f0, dtype=np.float64).reshape(nz, ny, nx).copy()
        a[:, 0, :] = hb.reshape(nz, nx)
        a[:, ny - 1, :] = ht.reshape(nz, nx)
        c = np.fft.fft(np.fft.rfft(a, axis=2), axis=0) * plan.norm
        u = np.zeros((ny, 2, Mm))
        v = np.zeros((ny, 2, Mm))
        for m in range(Mm):
            kz, kx = divmod(m, nxh)
            fm = np.ascontiguousarray(np.stack([c[kz, :, kx].real, c[kz, :, kx].imag], axis=1))       # (ny, 2) == Fortran f(2, ny)
            bcs = np.ascontiguousarray(np.stack([fm[0], fm[ny - 1]]))                                 # (2 walls, 2 lines)
            um, vm = R.ode2(2 if sing[m] else 1, lam[m], fm, bcs)
            u[:, :, m], v[:, :, m] = um, vm

        def back(w):
            cc = (w[:, 0, :] + 1j * w[:, 1, :]).reshape(ny, nz, nxh).transpose(1, 0, 2)
            return (np.fft.irfft(np.fft.ifft(cc, axis=0) * nz, n=nx, axis=2) * nx).reshape(-1)
        res["poisson_%s_u" % tag], res["poisson_%s_v" % tag] = u, v
        res["poisson_%s_p" % tag], res["poisson_%s_dpdy" % tag] = back(u), back(v)
        res["poisson_%s_f" % tag] = np.array([np.abs(f0).max()])
    np.savez(out, **res)


def rel(a, b):
    s = np.abs(b).max()
    return float(np.abs(a - b).max() / (s if s > 0 else 1.0))


def main():
    if not all(os.path.exists(p) for p in LIBS.values()):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "fma"], check=True)
    tmp = tempfile.mkdtemp()
    outs = {}
    for tag, lib in LIBS.items():
        outs[tag] = os.path.join(tmp, tag + ".npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", outs[tag]], check=True, env=dict(os.environ, TLAB_REF_LIB=lib))
    A, B = np.load(outs["fma"]), np.load(outs["plain"])
    fix, table = {}, {}
    for k in B.files:
        if k.startswith("poisson") and (k.endswith("_u") or k.endswith("_v") or k.endswith("_f")):
            continue
        fix["diff_" + k] = np.array(rel(A[k], B[k]))
        table[k] = float(fix["diff_" + k])
    nx, ny, nz = POISSON_GRID
    nxh = nx // 2 + 1
    for tag in ("first", "projected"):
        for w, name in (("u", "p"), ("v", "dpdy")):
            a, b = A["poisson_%s_%s" % (tag, w)], B["poisson_%s_%s" % (tag, w)]
            per_mode = np.abs(a - b).max(axis=(0, 1)) / np.abs(b).max()             # relative to the largest coefficient of the whole field
            worst = np.argsort(per_mode)[::-1][:4]
            fix["modes_%s_%s" % (tag, name)] = np.array([[m // nxh, m % nxh, per_mode[m]] for m in worst])
        fix["forcing_%s" % tag] = B["poisson_%s_f" % tag]
        fix["pressure_%s" % tag] = np.array([np.abs(B["poisson_%s_p" % tag]).max()])
    ver = subprocess.run(["amdflang", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    fix["_meta"] = np.array(json.dumps({"compiler": ver, "plain": "-O2 (x86-64 baseline: no fused multiply-add)", "fma": "-O2 -march=haswell (vfmadd contraction)",
                                        "line_lengths": LINE_LENGTHS, "poisson_grid": POISSON_GRID,
                                        "what": "max|fma build - plain build| / max|plain build| of the reference's own routines on identical inputs"}))
    out = os.path.join(ROOT, "tests", "golden", "ref_fma_scatter.npz")
    np.savez(out, **fix)
    for k in sorted(table):
        print("%-34s %.2e" % (k, table[k]))
    for tag in ("first", "projected"):
        print("forcing %-10s max|f| %.2e  max|p| %.2e   worst modes (kz, kx, diff) p: %s" % (tag, fix["forcing_%s" % tag][0], fix["pressure_%s" % tag][0],
              ", ".join("(%d,%d) %.1e" % (r[0], r[1], r[2]) for r in fix["modes_%s_p" % tag])))
    print("wrote", out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2])
    else:
        main()
