"""CPU tests of oracle/tlab_cpu.c, the C + OpenMP restatement that bench.py times as `cpu_baseline`: against the golden vectors generated from
the reference's own Fortran (tests/golden/) and against the numpy oracle, <= 1e-14 (the two differ at most by the last bit of a different
summation order inside the FFTs; everything else is the same operation sequence).

SECOND CHECKER of the transcribed routines.  OPR_Burgers_* (physics/opr_burgers.f90), OPR_Poisson_FourierXZ_Factorize (operators/opr_elliptic.f90)
and the RHS cannot be compiled from the reference in this image (opr_fourier.f90 needs fftw3.f03), so the numpy oracle's versions of them are pinned
through their compiled parts plus a reading of their data flow (oracle/ref_driver*.f90).  tlab_cpu.c is an independent second reading of the same
Fortran -- another language, its own transposes, sweeps, Stockham FFT and per-mode loop -- so test_operators_match_golden (OPR_Burgers against the
fixtures the reference's own solvers made), test_poisson_matches_numpy_oracle and test_substeps_match_numpy_oracle below check one transcription
against the other: a misreading of opr_burgers.f90 / opr_elliptic.f90 / rhs_global_incompressible_1.f90 would have to be made twice to pass."""
import os

import numpy as np
import pytest
from conftest import golden_files, rel_err
from oracle import tlab_oracle as O
from oracle import tlab_oracle_poisson as OP

TOL = 1e-14
C = pytest.importorskip("oracle.tlab_cpu")
pytestmark = pytest.mark.skipif(not os.path.exists(C.lib_path()), reason="oracle/libtlab_cpu.so not built (make -C oracle cpu)")


def plans_from_golden(g):
    spec = {1: (g["x"], True, True), 2: (g["y"], False, bool(g["yuniform"])), 3: (g["z"], True, True)}
    return {d: O.FdmPlan(n, p, u, int(g["mode1"]), int(g["mode2"])) for d, (n, p, u) in spec.items()}


@pytest.mark.parametrize("path", [p for p in golden_files("derivs_") if "c2n6" not in p and "penta" not in p])
def test_operators_match_golden(path):
    g = np.load(path)
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    plans = plans_from_golden(g)
    u, v, visc = g["u"], g["v"], float(g["visc"])
    for d in (1, 2, 3):
        t = C.FdmTables(plans[d])
        for ibc in ((0, 1, 2, 3) if d == 2 else (0,)):
            for typ in (O.OPR_P1, O.OPR_P2, O.OPR_P2_P1):
                r, t1 = C.opr_partial(d, typ, nx, ny, nz, ibc, t, u)
                assert rel_err(r, g["partial_d%d_t%d_bc%d" % (d, typ, ibc)]) <= TOL, (d, typ, ibc)
                if typ == O.OPR_P2_P1:
                    assert rel_err(t1, g["partial_d%d_t%d_bc%d_tmp1" % (d, typ, ibc)]) <= TOL
            r = C.opr_burgers(d, nx, ny, nz, ibc, t, visc, u, v)
            assert rel_err(r, g["burgers_d%d_bc%d" % (d, ibc)]) <= TOL, (d, ibc)


def test_transpose_bit_exact():
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (70, 130))           # Fortran a(130, 70): spans the 64 x 64 blocking
    b = np.empty((130, 70))
    C.load().tlabcpu_transpose(a.ctypes.data, 130, 70, b.ctypes.data)
    assert np.array_equal(b, a.T)


@pytest.mark.parametrize("nx,ny,nz,stretch", [(32, 40, 16, True), (16, 24, 1, True), (64, 33, 8, False), (24, 20, 10, True)])
def test_poisson_matches_numpy_oracle(nx, ny, nz, stretch):
    """power-of-two and other lengths (plain DFT), 2-D guard, all four singular modes"""
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / max(nz, 1) * 2 * np.pi if nz > 1 else np.zeros(1)
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5)) * 2.0 if stretch else np.arange(ny) / (ny - 1.0) * 2.0
    g = [O.FdmPlan(x, True, True), O.FdmPlan(y, False, not stretch), O.FdmPlan(z, True, True)]
    rng = np.random.default_rng(nx + ny + nz)
    N = nx * ny * nz
    i = np.arange(N)
    f = np.sin(0.3 * (i % nx)) * np.cos(0.07 * (i // nx)) + 0.2 * rng.uniform(-1, 1, N)
    hb, ht = rng.uniform(-1, 1, (nz, nx)), rng.uniform(-1, 1, (nz, nx))
    p_ref, d_ref = OP.opr_poisson_fxz(OP.PoissonPlan(g[0], g[1], g[2], nx, ny, nz), f, hb, ht)
    P = C.CpuPoisson(g[0], g[1], g[2], nx, ny, nz)
    p, dpdy = C.opr_poisson(P, f, hb, ht)
    assert rel_err(p, p_ref) <= 1e-13 and rel_err(dpdy, d_ref) <= 1e-13          # FFT summation order: ~1e-15 relative on the spectrum


@pytest.mark.parametrize("nx,ny,nz,stretch,nscal", [(32, 40, 16, True, 1), (64, 32, 32, False, 2)])
def test_substeps_match_numpy_oracle(nx, ny, nz, stretch, nscal):
    from oracle.tlab_oracle_rhs import DnsOracle
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5)) if stretch else np.arange(ny) / (ny - 1.0)
    sc = (0.7, 1.3)[:nscal]
    o = DnsOracle(x, y, z, nscal=nscal, visc=1.0 / 800.0, schmidt=sc, yuniform=not stretch)
    c = C.CpuDnsDriver(x, y, z, nscal=nscal, visc=1.0 / 800.0, schmidt=sc, yuniform=not stretch)
    rng = np.random.default_rng(7)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    for i in range(3 + nscal):
        a = ((np.sin(np.pi * X + i) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel()
        if i < 3:
            o.q[i] = a.copy(); c.q[i][:] = a
        else:
            o.s[i - 3] = a.copy(); c.s[i - 3][:] = a
    # two independent CPU implementations of the same algorithm differ only in the summation order inside their FFTs (1e-15 on the spectrum);
    # the projection amplifies that -- bound = max(1e-12, 2 x the oracle's own one-ulp scatter) (tests/scatter.py)
    from scatter import substep_scatter, bound
    sched = [(2e-3 / 3, -5.0 / 9.0, True), (2e-3 * 15 / 16, -153.0 / 128.0, True)]
    B, S = substep_scatter(lambda: DnsOracle(x, y, z, nscal=nscal, visc=1.0 / 800.0, schmidt=sc, yuniform=not stretch), o.q, o.s, sched, nsamples=3)
    for k, (dte, kco, scale) in enumerate(sched):
        c.time_substep(dte, kco, scale)
        for name in ("q", "hq", "s", "hs"):
            for i, (a, b) in enumerate(zip(getattr(c, name), B[k][name])):
                assert rel_err(a, b) <= bound(S[k][name][i]), (k, name, i, rel_err(a, b), S[k][name][i])
    assert C.load().tlabcpu_num_threads() >= 1


def test_port_as_the_checker_of_the_large_gpu_cases():
    """tests/scatter.py::cpu_port_factory -- what the GPU suite uses instead of the numpy oracle from 4e6 points on -- driven exactly as there (one driver
    per grid, handed out again with zeroed tendencies, new_step flags) and held to the numpy oracle through the same substep_scatter interface."""
    from oracle.tlab_oracle_rhs import DnsOracle
    from scatter import substep_scatter, bound, cpu_port_factory
    nx, ny, nz, nscal = 32, 40, 16, 2
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    sc = (0.7, 1.3)
    rng = np.random.default_rng(11)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    f = [((np.sin(np.pi * X + i) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for i in range(5)]
    sched = [(2e-3 / 3, -5.0 / 9.0, True, True), (2e-3 * 15 / 16, -153.0 / 128.0, True, False), (2e-3 * 8 / 15, 1.0, False, False), (2e-3 / 3, -5.0 / 9.0, True, True)]
    make = cpu_port_factory(x, y, z, nscal, 1.0 / 800.0, sc, False)
    Bc, Sc = substep_scatter(make, f[:3], f[3:], sched, nsamples=1)
    Bc2, _ = substep_scatter(make, f[:3], f[3:], sched, nsamples=1)          # the re-used driver starts from the same state
    Bn, Sn = substep_scatter(lambda: DnsOracle(x, y, z, nscal=nscal, visc=1.0 / 800.0, schmidt=sc, yuniform=False), f[:3], f[3:], sched, nsamples=2)
    for k in range(len(sched)):
        for name in ("q", "hq", "s", "hs"):
            for i, (a, b) in enumerate(zip(Bc[k][name], Bn[k][name])):
                assert np.array_equal(a, Bc2[k][name][i])
                assert rel_err(a, b) <= bound(Sn[k][name][i]), (k, name, i, rel_err(a, b), Sn[k][name][i])
