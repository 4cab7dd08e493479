"""[Staggering] StaggerHorizontalPressure (examples Case92, Case93): the interpolatory operators OPR_P0_INT_VP/PV, OPR_P1_INT_VP/PV
(opr_partial.f90:213-227, fdm_interpol.f90, fdm_compact_int.f90) and the staggered branch of RHS_GLOBAL_INCOMPRESSIBLE_1
(rhs_global_incompressible_1.f90:216-226, 266-273, 307-317) with its single singular Poisson mode (opr_elliptic.f90).

CPU: the oracle restatement against the reference's own outputs (tests/golden/stagger.npz, made by oracle/_ref), and the tables of the
library's plan generator.  GPU: the device operators and the staggered substep through the C ABI."""
import os
import numpy as np
import pytest
from conftest import rel_err, ROOT
import tlab_amd as T
from oracle import tlab_oracle as O

G = np.load(os.path.join(ROOT, "tests", "golden", "stagger.npz"))
TYPES = (T.OPR_P1_INT_VP, T.OPR_P1_INT_PV, T.OPR_P0_INT_VP, T.OPR_P0_INT_PV)


def device_partial(d, t, g, nx, ny, nz, u):
    import torch
    r = torch.empty_like(u)
    (T.OPR_Partial_X, T.OPR_Partial_Y, T.OPR_Partial_Z)[d - 1](t, nx, ny, nz, 0, g, u, r)
    return r


def plans(mod):
    y = np.arange(8) / 7.0        # (the fixture's 6 y planes only batch the x / z lines; the library's plans need >= 8 nodes)
    return (mod.FdmPlan(G["x"], True, True, stagger=True), mod.FdmPlan(y, False, True), mod.FdmPlan(G["z"], True, True, stagger=True))


def test_type_constants_are_the_reference_values():
    assert TYPES == (5, 6, 7, 8)      # TLab_Constants: OPR_P1_INT_VP .. OPR_P0_INT_PV


def test_oracle_tables_and_operators_equal_the_reference():
    gx, gy, gz = plans(O)
    nz, ny, nx = G["u"].shape
    for d, g in ((1, gx), (3, gz)):
        assert np.array_equal(g.intl.lu0i, G["plan%d_lu0i" % d]) and np.array_equal(g.intl.lu1i, G["plan%d_lu1i" % d])
        assert np.array_equal(g.der1.mwn, G["plan%d_mwn1" % d])
        for t in TYPES:
            r = O.opr_partial(d, t, nx, ny, nz, 0, g, G["u"])[0]
            assert np.array_equal(np.asarray(r).reshape(nz, ny, nx), G["d%d_t%d" % (d, t)]), (d, t)


def test_library_plan_generator_tables():
    gx, gy, gz = plans(T)
    for d, g in ((1, gx), (3, gz)):
        assert g.info(7) == 1
        assert rel_err(g.table("lu0i"), G["plan%d_lu0i" % d]) <= 1e-14
        assert rel_err(g.table("lu1i"), G["plan%d_lu1i" % d]) <= 1e-14
        assert rel_err(g.table("mwn1"), G["plan%d_mwn1" % d]) <= 1e-14
    assert gy.info(7) == 0
    # non-periodic directions ignore the switch (fdm.f90:236: only g%periodic), and type 5-8 on a plan without tables is an error
    p = T.FdmPlan(np.arange(8) / 7.0, False, True, stagger=True)
    assert p.info(7) == 0


def test_interpolation_is_a_half_cell_shift_of_a_resolved_wave():
    """property: P0_INT_VP of cos(k x) is cos(k (x + h/2)) up to the scheme's sixth-order error, and PV undoes VP's shift"""
    n = 64
    x = np.arange(n) / n * 2 * np.pi
    g = O.FdmPlan(x, True, True, stagger=True)
    u = np.cos(2 * x).reshape(1, 1, n)
    h = x[1] - x[0]
    a = np.asarray(O.opr_partial(1, 7, n, 1, 1, 0, g, u)[0]).reshape(n)
    assert np.abs(a - np.cos(2 * (x + h / 2))).max() < 1e-7
    b = np.asarray(O.opr_partial(1, 8, n, 1, 1, 0, g, a)[0]).reshape(n)
    assert np.abs(b - np.cos(2 * x)).max() < 1e-7
    d = np.asarray(O.opr_partial(1, 5, n, 1, 1, 0, g, u)[0]).reshape(n)
    assert np.abs(d + 2 * np.sin(2 * (x + h / 2))).max() < 1e-6


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    T.init(0)
    return T


@pytest.mark.gpu_extra
@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 3])
@pytest.mark.parametrize("t", TYPES)
def test_device_interpolatory_operators_vs_fixture(dev, d, t):
    import torch
    gx, gy, gz = plans(T)
    nz, ny, nx = G["u"].shape
    u = torch.from_numpy(G["u"].copy()).cuda().reshape(-1)
    r = device_partial(d, t, (gx, gy, gz)[d - 1], nx, ny, nz, u)
    assert rel_err(r.cpu().numpy().reshape(nz, ny, nx), G["d%d_t%d" % (d, t)]) <= 1e-13, (d, t)


@pytest.mark.gpu_extra
@pytest.mark.gpu
@pytest.mark.parametrize("n", [256, 1024])
def test_device_interpolatory_operators_long_lines(dev, n):
    import torch
    rng = np.random.default_rng(n)
    x = np.arange(n) / n * 2 * np.pi
    g, o = T.FdmPlan(x, True, True, stagger=True), O.FdmPlan(x, True, True, stagger=True)
    for d, shape in ((1, (3, 4, n)), (3, (n, 4, 3))):
        u = rng.uniform(-1, 1, shape)
        nz, ny, nx = shape
        for t in TYPES:
            r = device_partial(d, t, g, nx, ny, nz, torch.from_numpy(u).cuda().reshape(-1))
            e = O.opr_partial(d, t, nx, ny, nz, 0, o, u)[0]
            assert rel_err(r.cpu().numpy().reshape(shape), np.asarray(e).reshape(shape)) <= 1e-12, (d, t)


@pytest.mark.gpu_extra
@pytest.mark.gpu
def test_interpolatory_type_without_tables_is_refused(dev):
    import torch
    n = 32
    g = T.FdmPlan(np.arange(n) / n, True, True)
    with pytest.raises(T.TlabError):
        device_partial(1, T.OPR_P0_INT_VP, g, n, 2, 2, torch.zeros(n * 4, dtype=torch.float64, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_staggered_substeps_vs_oracle(dev, fuse):
    """Two Runge-Kutta substeps with the pressure on the horizontally staggered grid: forcing interpolated VP in x and z, one singular
    Fourier mode, gradient interpolated back PV; within the scatter bound of the oracle under 1-ulp input noise (tests/scatter.py)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    from test_gpu_rhs import grids, init_fields, oracle_substeps, check_state, REF_HYPER
    nx, ny, nz = 64, 40, 32
    x, y, z = grids(nx, ny, nz, True)
    visc, sc = 1.0 / 800.0, (0.7,)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 23)
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER, stagger=True)
    d.set_fusion(fuse)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    sched = [(2e-3 * d.kdt[k], d.kco[k], True) for k in range(2)]
    B, S = oracle_substeps(("stagger",), lambda: DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, stagger=True),
                           q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, tag="stagger")
