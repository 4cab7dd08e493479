"""CPU tests: the numpy oracle of the Poisson path is pinned against the reference-generated golden vectors
(per-mode FDM_Int1 systems, solves and OPR_ODE2_Factorize_NN/_NN_Sing), against oracle/_ref when present, and through the
discrete identity  div(grad p) = f  (P1 o P1, as vpoisson.f90 / SURVEY.md 4.4)."""
import numpy as np
import pytest
from conftest import golden_files, rel_err
from oracle import tlab_oracle as O
from oracle import tlab_oracle_poisson as OP

TOL = 1e-13


@pytest.mark.parametrize("path", golden_files("poisson_modes_"))
def test_int1_and_ode2_match_golden(path):
    g = np.load(path)
    y = g["y"]; n = y.shape[0]
    # mode1: the first-derivative scheme whose integral operators these are (absent in the round-1 fixtures: CompactJacobian6); 5 = CompactJacobian6Penta
    # -> 7-diagonal systems (HEPTADFS / HEPTADSS, MatMul_5d), 4 = CompactJacobian4 -> tridiagonal ones (fdm_integral.f90:75-83, 249-263)
    gy = O.FdmPlan(y, False, bool(g["uniform"]), mode1=int(g["mode1"]) if "mode1" in g.files else O.FDM_COM6_JACOBIAN)
    for il, lam in enumerate(g["lams"]):
        for ibc, sgn in ((1, 1.0), (2, -1.0)):
            p = OP.int1_create_system(gy.der1, sgn * lam, ibc)
            for k, a in (("lhs", p.lhs[:, :, 0]), ("rhs", p.rhs), ("rhs_b", p.rhs_b[:, :, 0]), ("rhs_t", p.rhs_t[:, :, 0])):
                assert rel_err(a, g["sys_l%d_bc%d_%s" % (il, ibc, k)]) <= TOL, (il, ibc, k)
            p = OP.int1_initialize(gy.der1, sgn * lam, ibc)
            assert rel_err(p.lhs[:, :, 0], g["lu_l%d_bc%d_lhs" % (il, ibc)]) <= TOL
            f = g["int1_l%d_bc%d_f" % (il, ibc)]
            r = g["int1_l%d_bc%d_res0" % (il, ibc)].copy().reshape(n, 3, 1)
            du = OP.int1_solve(p, p.rhs, f.reshape(n, 3, 1), r, want_du=True)
            assert rel_err(r[:, :, 0], g["int1_l%d_bc%d_res" % (il, ibc)]) <= TOL
            assert rel_err(du[:, 0], g["int1_l%d_bc%d_du" % (il, ibc)]) <= TOL
        fmin = OP.int1_initialize(gy.der1, lam, 1)
        fmax = OP.int1_initialize(gy.der1, -lam, 2)
        fn = OP.ode2_factorize_nn if int(g["ode2_l%d_type" % il]) == 1 else OP.ode2_factorize_nn_sing
        u, v = fn(fmin, fmax, g["ode2_l%d_f" % il].reshape(n, 2, 1).copy(), g["ode2_l%d_bcs" % il].reshape(2, 2, 1))
        assert rel_err(u[:, :, 0], g["ode2_l%d_u" % il]) <= 1e-12, il
        assert rel_err(v[:, :, 0], g["ode2_l%d_v" % il]) <= 1e-12, il
        fn = OP.ode2_factorize_dd if int(g["ode2dd_l%d_type" % il]) == 3 else OP.ode2_factorize_dd_sing
        u, v = fn(fmin, fmax, g["ode2dd_l%d_f" % il].reshape(n, 2, 1).copy(), g["ode2dd_l%d_bcs" % il].reshape(2, 2, 1))
        assert rel_err(u[:, :, 0], g["ode2dd_l%d_u" % il]) <= 1e-12, il
        assert rel_err(v[:, :, 0], g["ode2dd_l%d_v" % il]) <= 1e-12, il


def _poisson_setup(nx, ny, nz, seed=0):
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * 2 * np.pi
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5)) * 2.0
    gx, gy, gz = O.FdmPlan(x, True, True), O.FdmPlan(y, False, False), O.FdmPlan(z, True, True)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    rng = np.random.default_rng(seed)
    phi = (np.sin(X) * np.cos(2 * Z) * np.exp(0.5 * Y) + np.cos(3 * X + 1) * Y ** 2 + 0.3 * np.sin(2 * Z) * np.cos(2 * Y)
           + 0.01 * rng.uniform(-1, 1, X.shape)).ravel()
    return gx, gy, gz, phi


@pytest.mark.parametrize("nx,ny,nz", [(32, 40, 16), (16, 24, 1)])
def test_poisson_dirichlet_discrete_identity(nx, ny, nz):
    """ibc = BCS_DD (OPR_ODE2_Factorize_DD / _DD_Sing per mode): p takes the given wall values and div(grad p) = f with P1 o P1."""
    gx, gy, gz, phi = _poisson_setup(nx, ny, nz, seed=3)

    def P1(d, g, u):
        return O.opr_partial(d, 1, nx, ny, nz, 0, g, u)[0]

    dphidy = P1(2, gy, phi)
    f = P1(1, gx, P1(1, gx, phi)) + P1(2, gy, dphidy) + (P1(3, gz, P1(3, gz, phi)) if nz > 1 else 0.0)
    p3 = phi.reshape(nz, ny, nx)
    plan = OP.PoissonPlan(gx, gy, gz, nx, ny, nz)
    p, dpdy = OP.opr_poisson_fxz(plan, f, p3[:, 0, :].copy(), p3[:, ny - 1, :].copy(), ibc=O.BCS_DD)
    assert rel_err(p, phi) <= 1e-9                      # Dirichlet data pin the solution itself
    assert rel_err(dpdy, dphidy) <= 1e-8


@pytest.mark.parametrize("nx,ny,nz", [(32, 40, 16), (16, 24, 1), (64, 33, 8)])
def test_poisson_discrete_identity(nx, ny, nz):
    gx, gy, gz, phi = _poisson_setup(nx, ny, nz)

    def P1(d, g, u):
        return O.opr_partial(d, 1, nx, ny, nz, 0, g, u)[0]

    dphidy = P1(2, gy, phi)
    f = P1(1, gx, P1(1, gx, phi)) + P1(2, gy, dphidy) + P1(3, gz, P1(3, gz, phi))
    d3 = dphidy.reshape(nz, ny, nx)
    plan = OP.PoissonPlan(gx, gy, gz, nx, ny, nz)
    p, dpdy = OP.opr_poisson_fxz(plan, f, d3[:, 0, :].copy(), d3[:, -1, :].copy())
    assert rel_err(dpdy, dphidy) <= 1e-12
    res = P1(1, gx, P1(1, gx, p)) + P1(2, gy, dpdy) + P1(3, gz, P1(3, gz, p)) - f
    assert np.abs(res).max() / np.abs(f).max() <= 1e-12
    # the reference pins p = 0 at the bottom wall for the mean mode (opr_odes.f90:179-180)
    assert abs(p.reshape(nz, ny, nx)[:, 0, :].mean()) <= 1e-12 * np.abs(p).max()


@pytest.mark.parametrize("nx,ny,nz,ibc,alpha", [(32, 40, 16, O.BCS_NN, -7.5), (16, 24, 1, O.BCS_DD, -120.0), (16, 48, 8, O.BCS_DD, -0.5)])
def test_helmholtz_factorized_discrete_identity(nx, ny, nz, ibc, alpha):
    """OPR_Helmholtz_FourierXZ_Factorize: with f = (P1 o P1 Laplacian + alpha) phi and phi's own wall data the solver returns phi (alpha < 0: no
    null space, so the data pin the solution for both boundary types)."""
    gx, gy, gz, phi = _poisson_setup(nx, ny, nz, seed=5)

    def P1(d, g, u):
        return O.opr_partial(d, 1, nx, ny, nz, 0, g, u)[0]

    dphidy = P1(2, gy, phi)
    f = P1(1, gx, P1(1, gx, phi)) + P1(2, gy, dphidy) + (P1(3, gz, P1(3, gz, phi)) if nz > 1 else 0.0) + alpha * phi
    w = (dphidy if ibc == O.BCS_NN else phi).reshape(nz, ny, nx)
    plan = OP.PoissonPlan(gx, gy, gz, nx, ny, nz)
    a = OP.opr_helmholtz_fxz_factorize(plan, f, w[:, 0, :].copy(), w[:, ny - 1, :].copy(), ibc, alpha)
    assert rel_err(a, phi) <= 1e-10


# ---------------------------------------------------------------------------------------------------------------------------------
# DIRECT elliptic solver (EllipticOrder = CompactDirect6): FDM_Int2_* and OPR_Poisson_FourierXZ_Direct
# ---------------------------------------------------------------------------------------------------------------------------------
def _direct_plan_from(g, prefix="tab_", nodes_key="y"):
    tab = {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}
    tab["nodes"] = g[nodes_key]
    return O.FdmPlan.from_tables(tab)


@pytest.mark.parametrize("path", golden_files("poisson_direct_modes_"))
def test_int2_matches_golden(path):
    g = np.load(path)
    y = g["y"]; n = y.shape[0]
    gy = _direct_plan_from(g)
    for il, lam in enumerate(g["lams"]):
        for ibc in (0, 1, 2, 3):
            if "lu_l%d_bc%d_lhs" % (il, ibc) not in g.files:
                continue
            p = OP.int2_initialize(gy.der2, y, lam, ibc)
            for k, a in (("lhs", p.lhs[:, :, 0]), ("rhs", p.rhs), ("rhs_b", p.rhs_b), ("rhs_t", p.rhs_t)):
                assert rel_err(a, g["lu_l%d_bc%d_%s" % (il, ibc, k)]) <= TOL, (il, ibc, k)
            f = g["int2_l%d_bc%d_f" % (il, ibc)]
            r = g["int2_l%d_bc%d_res0" % (il, ibc)].copy().reshape(n, 2, 1)
            OP.int2_solve(p, p.rhs, f.reshape(n, 2, 1), r)
            assert rel_err(r[:, :, 0], g["int2_l%d_bc%d_res" % (il, ibc)]) <= TOL, (il, ibc)


@pytest.mark.parametrize("ny,nx,nz,ibc", [(24, 16, 8, 3), (64, 16, 1, 3), (64, 8, 8, 0), (24, 8, 8, 1), (24, 8, 8, 2)])
def test_poisson_direct_discrete_identity(ny, nx, nz, ibc):
    """(d2/dx2 + d2/dy2 + d2/dz2) p = f at the interior rows with the second-derivative operators of the plans (direct in y), and the
    boundary rows carry the data: Dirichlet value, or -- Neumann -- the 4th-order biased derivative of fdm_integral.f90:436-514."""
    g = np.load(golden_files("direct_y")[0])
    tab = {k[len("ny%d_" % ny):]: g[k] for k in g.files if k.startswith("ny%d_" % ny)}
    gy = O.FdmPlan.from_tables(tab)
    y = gy.nodes
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * 2 * np.pi
    gx, gz = O.FdmPlan(x, True, True), O.FdmPlan(z, True, True)
    rng = np.random.default_rng(ny + ibc)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    f = (np.sin(X) * np.cos(Z) * np.exp(Y) + 0.2 * rng.uniform(-1, 1, X.shape)).ravel()
    hb, ht = rng.uniform(-1, 1, (nz, nx)), rng.uniform(-1, 1, (nz, nx))
    plan = OP.PoissonDirectPlan(gx, gy, gz, nx, ny, nz)
    p, dpdy = OP.opr_poisson_fxz_direct(plan, f, hb, ht, ibc)

    def P2(d, gg, u):
        return O.opr_partial(d, 2, nx, ny, nz, 0, gg, u)[0]

    lap = P2(1, gx, p) + P2(2, gy, p) + (P2(3, gz, p) if nz > 1 else 0.0)
    L3, F3, P3 = lap.reshape(nz, ny, nx), f.reshape(nz, ny, nx), p.reshape(nz, ny, nx)
    if ibc == 3:      # the mean mode is solved with p = 0 at the bottom instead of its Neumann datum: compare without the xz-mean
        L3 = L3 - L3.mean(axis=(0, 2), keepdims=True)
        F3 = F3 - F3.mean(axis=(0, 2), keepdims=True)
    assert rel_err(L3[:, 3:ny - 3, :], F3[:, 3:ny - 3, :]) <= 1e-9
    if ibc in (0, 2):
        assert rel_err(P3[:, 0, :], hb) <= 1e-12
    if ibc in (0, 1):
        assert rel_err(P3[:, ny - 1, :], ht) <= 1e-12
    # Neumann ends: p'_1 = b1 p1 + b2 p2 + b3 p3 + b4 p4 + a2 p''_2 with p''_2 = (f - d2p/dx2 - d2p/dz2)_2 (fdm_integral.f90:436-514)
    rest = f.reshape(nz, ny, nx) - (P2(1, gx, p) + (P2(3, gz, p) if nz > 1 else 0.0)).reshape(nz, ny, nx)
    if ibc in (1, 3):
        c = OP.coef_c1n4_biased(y, 1)
        got = c[0] * P3[:, 0] + c[1] * P3[:, 1] + c[2] * P3[:, 2] + c[3] * P3[:, 3] + c[4] * rest[:, 1]
        want = hb
        if ibc == 3:      # the mean mode has p = 0 at the bottom instead (compatibility constraint, opr_elliptic.f90:420-421)
            got, want = got - got.mean(), want - want.mean()
            assert abs(P3[:, 0].mean()) <= 1e-13 * np.abs(P3).max()
        assert rel_err(got, want) <= 1e-10
    if ibc in (2, 3):
        c = OP.coef_c1n4_biased(y, ny, backwards=True)
        got = c[0] * P3[:, ny - 1] + c[1] * P3[:, ny - 2] + c[2] * P3[:, ny - 3] + c[3] * P3[:, ny - 4] + c[4] * rest[:, ny - 2]
        assert rel_err(got, ht) <= 1e-10
    assert np.isfinite(dpdy).all()


@pytest.mark.parametrize("ibc", [0, 1, 2, 3])
def test_helmholtz_direct_discrete_identity(ibc):
    """(lap + alpha) a = f at the interior rows with the second-derivative operators of the plans (OPR_Helmholtz_FourierXZ_Direct)."""
    ny, nx, nz, alpha = 64, 16, 8, -37.5
    g = np.load(golden_files("direct_y")[0])
    tab = {k[len("ny%d_" % ny):]: g[k] for k in g.files if k.startswith("ny%d_" % ny)}
    gy = O.FdmPlan.from_tables(tab)
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * 2 * np.pi
    gx, gz = O.FdmPlan(x, True, True), O.FdmPlan(z, True, True)
    rng = np.random.default_rng(ibc)
    f = rng.uniform(-1, 1, nx * ny * nz)
    hb, ht = rng.uniform(-1, 1, (nz, nx)), rng.uniform(-1, 1, (nz, nx))
    plan = OP.PoissonDirectPlan(gx, gy, gz, nx, ny, nz)
    a = OP.opr_helmholtz_fxz_direct(plan, f, hb, ht, ibc, alpha)
    lap = sum(O.opr_partial(d, 2, nx, ny, nz, 0, gg, a)[0] for d, gg in ((1, gx), (2, gy), (3, gz))) + alpha * a
    assert rel_err(lap.reshape(nz, ny, nx)[:, 3:ny - 3], f.reshape(nz, ny, nx)[:, 3:ny - 3]) <= 1e-9
    A3 = a.reshape(nz, ny, nx)
    if ibc in (0, 2):
        assert rel_err(A3[:, 0], hb) <= 1e-12
    if ibc in (0, 1):
        assert rel_err(A3[:, ny - 1], ht) <= 1e-12
