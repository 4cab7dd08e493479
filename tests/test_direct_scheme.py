"""SpaceOrder2 = CompactDirect6 in the non-periodic direction (fdm_comx_direct.f90 + MatMul_5d).  The coefficient
tables are the reference's own (tests/golden/direct_y.npz, made by FDM_CreatePlan through oracle/_ref); what is tested is the operator
path on top of them: oracle (MatMul_5d restatement) on CPU, device kernels with per-row right-hand sides on the GPU."""
import os

import numpy as np

REF_HYPER = 0.1      # wall closure of the flang-built reference (defect 1); the driver classes default to the consistent 0.0
import pytest
from conftest import rel_err

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct_y.npz"))
KEYS = ("ndl1", "ndr1", "ndl2", "ndr2", "need_1der", "lhs1", "rhs1", "lu1", "rhs_b1", "rhs_t1", "mwn1", "lhs2", "rhs2", "lu2", "mwn2", "jac", "nodes")


def tables(ny):
    return {k: G["ny%d_%s" % (ny, k)] for k in KEYS}


def test_oracle_direct_scheme_matches_reference_outputs():
    from oracle import tlab_oracle as O
    nx, ny, nz = 16, 24, 8
    g = O.FdmPlan.from_tables(tables(ny), mode2=O.FDM_COM6_DIRECT)
    assert g.der2.direct and not g.der2.need_1der and int(G["x_ndr2"]) == 7
    u, v, visc = G["u"], G["v"], float(G["visc"])
    for t in (1, 2, 3):
        r, t1 = O.opr_partial(2, t, nx, ny, nz, 0, g, u)
        assert rel_err(r, G["partial_t%d" % t]) <= 1e-14
        if t == 3:
            assert rel_err(t1, G["partial_t3_tmp1"]) <= 1e-14
    assert rel_err(O.opr_burgers(2, nx, ny, nz, 0, g, visc, u, v)[0], G["burgers"]) <= 1e-14


G1 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct1_y.npz"))


def tables1(mode, ny):
    t = {k: G1["m%d_ny%d_%s" % (mode, ny, k)] for k in KEYS if k != "nodes"}
    t["nodes"] = G1["ny%d_nodes" % ny]
    return t


def direct1_cases(mode, ny):
    """(operator, type, ibc, expected[, expected tmp1]) of the fixture for one plan"""
    pre = "m%d_ny%d_" % (mode, ny)
    for ibc in (0, 1, 2, 3):
        for t in (1, 2, 3):
            k = pre + "partial_t%d_bc%d" % (t, ibc)
            if k in G1:
                yield "partial", t, ibc, G1[k], (G1[pre + "partial_t3_bc%d_tmp1" % ibc] if t == 3 else None)
        k = pre + "burgers_bc%d" % ibc
        if k in G1:
            yield "burgers", 0, ibc, G1[k], None


@pytest.mark.parametrize("mode", [17, 16])
@pytest.mark.parametrize("ny", [24, 72])
def test_oracle_direct_first_derivative_matches_reference_outputs(mode, ny):
    """SpaceOrder1 = CompactDirect4 / CompactDirect6: MatMul_3d / MatMul_5d with the Neumann rows (fdm_matmul.f90:70-121, 265-319)."""
    from oracle import tlab_oracle as O
    nx, nz = int(G1["nx"]), int(G1["nz"])
    g = O.FdmPlan.from_tables(tables1(mode, ny), mode1=mode, mode2=mode)
    assert g.der1.direct and g.der1.nb_diag == (3, 3 if mode == 17 else 5)
    u, v, visc = G1["ny%d_u" % ny], G1["ny%d_v" % ny], float(G1["visc"])
    for op, t, ibc, want, want1 in direct1_cases(mode, ny):
        if op == "partial":
            r, t1 = O.opr_partial(2, t, nx, ny, nz, ibc, g, u)
            assert rel_err(r, want) <= 1e-14, (t, ibc)
            if want1 is not None:
                assert rel_err(t1, want1) <= 1e-14, (t, ibc)
        else:
            assert rel_err(O.opr_burgers(2, nx, ny, nz, ibc, g, visc, u, v)[0], want) <= 1e-14, ibc


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [17, 16])
@pytest.mark.parametrize("ny", [24, 72])
def test_device_direct_first_derivative(mode, ny):
    """The device path of the same plans (tables from the host, tlab_fdm_plan_create_from_arrays + tlab_fdm_plan_set_modes; k_line1: one
    line per thread, per-row right-hand side) against the reference's outputs."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    nx, nz = int(G1["nx"]), int(G1["nz"])
    gp = T.FdmPlan.from_tables(tables1(mode, ny), periodic=False, scheme1=mode, scheme2=mode)
    u, v, visc = G1["ny%d_u" % ny], G1["ny%d_v" % ny], float(G1["visc"])
    du, dv = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    res, tmp = torch.empty_like(du), torch.empty_like(du)
    for op, t, ibc, want, want1 in direct1_cases(mode, ny):
        res.fill_(float("nan")); tmp.fill_(float("nan"))
        if op == "partial":
            T.OPR_Partial_Y(t, nx, ny, nz, ibc, gp, du, res, tmp)
            assert rel_err(res.cpu().numpy(), want) <= 1e-12, (t, ibc)
            if want1 is not None:
                assert rel_err(tmp.cpu().numpy(), want1) <= 1e-12, (t, ibc)
        else:
            T.OPR_Burgers_Y(T.OPR_B_U_IN, visc, nx, ny, nz, ibc, gp, du, dv, res, tmp)
            assert rel_err(res.cpu().numpy(), want) <= 1e-12, ibc


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [17, 16])
def test_device_direct_first_derivative_on_the_tile_kernels(mode):
    """128-point lines (register-tile kernel for OPR_P1, half-wave tiles for the fused forms) against the oracle on the reference's tables."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    from tlab_amd.lib import load
    from oracle import tlab_oracle as O
    T.init(0)
    nx, ny, nz = 48, 128, 3
    tab = tables1(mode, ny)
    gp = T.FdmPlan.from_tables(tab, periodic=False, scheme1=mode, scheme2=mode)
    go = O.FdmPlan.from_tables(tab, mode1=mode, mode2=mode)
    rng = np.random.default_rng(mode)
    u, v, visc = rng.uniform(-1, 1, nx * ny * nz), rng.uniform(-1, 1, nx * ny * nz), 1.0 / 300.0
    du, dv = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    res, tmp = torch.empty_like(du), torch.empty_like(du)
    for ibc in (0, 1, 2, 3):
        for t in (T.OPR_P1, T.OPR_P2_P1):
            res.fill_(float("nan")); tmp.fill_(float("nan"))
            T.OPR_Partial_Y(t, nx, ny, nz, ibc, gp, du, res, tmp)
            assert load().tlab_last_kernel_path() == 3
            r, t1 = O.opr_partial(2, t, nx, ny, nz, ibc, go, u)
            assert rel_err(res.cpu().numpy(), r) <= 1e-12, (t, ibc)
            if t == T.OPR_P2_P1:
                assert rel_err(tmp.cpu().numpy(), t1) <= 1e-12, (t, ibc)
        res.fill_(float("nan"))
        T.OPR_Burgers_Y(T.OPR_B_U_IN, visc, nx, ny, nz, ibc, gp, du, dv, res, tmp)
        assert rel_err(res.cpu().numpy(), O.opr_burgers(2, nx, ny, nz, ibc, go, visc, u, v)[0]) <= 1e-12, ibc


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nx,nz", [(24, 16, 8), (64, 64, 3), (128, 48, 2), (512, 32, 2)])
def test_device_direct_scheme(ny, nx, nz):
    """24: generic kernel against the reference's outputs; 64 / 128 / 512: register-tile and half-wave-tile kernels against the oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    from oracle import tlab_oracle as O
    T.init(0)
    tab = tables(ny)
    gp = T.FdmPlan.from_tables(tab, periodic=False, scheme1=T.FDM_COM6_JACOBIAN, scheme2=T.FDM_COM6_DIRECT)
    go = O.FdmPlan.from_tables(tab, mode2=O.FDM_COM6_DIRECT)
    if ny == 24:
        u, v, visc = G["u"], G["v"], float(G["visc"])
    else:
        rng = np.random.default_rng(ny)
        u, v, visc = rng.uniform(-1, 1, nx * ny * nz), rng.uniform(-1, 1, nx * ny * nz), 1.0 / 300.0
    du, dv = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    res, tmp = torch.empty_like(du), torch.empty_like(du)
    for t in (T.OPR_P1, T.OPR_P2, T.OPR_P2_P1):
        res.fill_(float("nan")); tmp.fill_(float("nan"))
        T.OPR_Partial_Y(t, nx, ny, nz, 0, gp, du, res, tmp)
        r, t1 = O.opr_partial(2, t, nx, ny, nz, 0, go, u)
        assert rel_err(res.cpu().numpy(), r) <= 1e-12, t
        if t == T.OPR_P2_P1:
            assert rel_err(tmp.cpu().numpy(), t1) <= 1e-12
        if ny == 24:
            assert rel_err(res.cpu().numpy(), G["partial_t%d" % t]) <= 1e-12
    res.fill_(float("nan"))
    T.OPR_Burgers_Y(T.OPR_B_U_IN, visc, nx, ny, nz, 0, gp, du, dv, res, tmp)
    assert rel_err(res.cpu().numpy(), O.opr_burgers(2, nx, ny, nz, 0, go, visc, u, v)[0]) <= 1e-12
    if ny == 24:
        assert rel_err(res.cpu().numpy(), G["burgers"]) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("elliptic_direct", [False, True])
def test_substep_with_direct_second_derivative_in_y(elliptic_direct):
    """The scheme set of examples/Case81-93 (SpaceOrder2 = CompactDirect6; with and without EllipticOrder = CompactDirect6, everything else
    default): full substeps against the oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    from tlab_amd.dns import Dns
    from oracle import tlab_oracle as O
    from oracle.tlab_oracle_rhs import DnsOracle
    T.init(0)
    nx, ny, nz = 256, 64, 32
    tab = tables(ny)
    x, y, z = np.arange(nx) / nx * 2.0, tab["nodes"], np.arange(nz) / nz
    gp = [T.FdmPlan(x, True, True), T.FdmPlan.from_tables(tab, False, T.FDM_COM6_JACOBIAN, T.FDM_COM6_DIRECT), T.FdmPlan(z, True, True)]
    go = [O.FdmPlan(x, True, True), O.FdmPlan.from_tables(tab, mode2=O.FDM_COM6_DIRECT), O.FdmPlan(z, True, True)]
    d = Dns(x, y, z, nscal=1, visc=1.0 / 800.0, schmidt=(0.7,), yuniform=False, plans=gp, gy_elliptic=gp[1] if elliptic_direct else None, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=1, visc=1.0 / 800.0, schmidt=(0.7,), yuniform=False, plans=go, gy_elliptic=go[1] if elliptic_direct else None)
    rng = np.random.default_rng(81)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * Y)
    for i in range(3):
        a = ((np.sin(np.pi * X + i) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel()
        d.q[i].copy_(torch.from_numpy(a)); o.q[i] = a.copy()
    a = (np.cos(np.pi * X) * Y + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()
    d.s[0].copy_(torch.from_numpy(a)); o.s[0] = a.copy()
    for k in range(2):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(2e-3 * d.kdt[k], d.kco[k], True)
        o.time_substep(2e-3 * d.kdt[k], d.kco[k], True)
        for i in range(3):
            assert rel_err(d.q[i].cpu().numpy(), o.q[i]) <= 1e-12, (k, i)
        assert rel_err(d.s[0].cpu().numpy(), o.s[0]) <= 1e-12


@pytest.mark.gpu
def test_jacobian_derivatives_with_direct_elliptic_solver():
    """[Main] EllipticOrder = CompactDirect6 on its own (the derivatives keep CompactJacobian6 / CompactJacobian6Hyper): the Poisson solver uses
    fdm_loc's direct second derivative while dp/dy and the divergence use the Jacobian first derivative (opr_elliptic.f90:107-124, :447-449)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    from tlab_amd.dns import Dns
    from oracle import tlab_oracle as O
    from oracle.tlab_oracle_rhs import DnsOracle
    from scatter import substep_scatter, bound
    T.init(0)
    nx, ny, nz = 256, 64, 32
    tab = tables(ny)
    x, y, z = np.arange(nx) / nx * 2.0, tab["nodes"], np.arange(nz) / nz
    gp = [T.FdmPlan(x, True, True), T.FdmPlan(y, False, False), T.FdmPlan(z, True, True)]
    ge = T.FdmPlan.from_tables(tab, False, T.FDM_COM6_JACOBIAN, T.FDM_COM6_DIRECT)
    d = Dns(x, y, z, nscal=1, visc=1.0 / 800.0, schmidt=(0.7,), yuniform=False, plans=gp, gy_elliptic=ge, hyper_bc1_ext=REF_HYPER)

    def make_oracle():
        go = [O.FdmPlan(x, True, True), O.FdmPlan(y, False, False), O.FdmPlan(z, True, True)]
        return DnsOracle(x, y, z, nscal=1, visc=1.0 / 800.0, schmidt=(0.7,), yuniform=False, plans=go, gy_elliptic=O.FdmPlan.from_tables(tab, mode2=O.FDM_COM6_DIRECT))
    rng = np.random.default_rng(82)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    q0 = [((np.sin(np.pi * X + i) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for i in range(3)]
    s0 = [(np.cos(np.pi * X) * Y + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()]
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    sched = [(2e-3 * d.kdt[k], d.kco[k], True) for k in range(2)]
    B, S = substep_scatter(make_oracle, q0, s0, sched, nsamples=1)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
    for i in range(3):
        assert rel_err(d.q[i].cpu().numpy(), B[1]["q"][i]) <= bound(S[1]["q"][i]), (i, rel_err(d.q[i].cpu().numpy(), B[1]["q"][i]), S[1]["q"][i])
    assert rel_err(d.s[0].cpu().numpy(), B[1]["s"][0]) <= bound(S[1]["s"][0])


@pytest.mark.gpu
def test_direct_scheme_along_x_takes_the_generic_kernel():
    """A non-periodic, stretched x with a direct second derivative (n = 512 would otherwise select the wave-per-line kernel, which only
    knows the constant stencils of the Jacobian schemes)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    from oracle import tlab_oracle as O
    from tlab_amd.lib import load
    T.init(0)
    nx, ny, nz = 512, 4, 3
    tab = tables(nx)
    gp = T.FdmPlan.from_tables(tab, periodic=False, scheme1=T.FDM_COM6_JACOBIAN, scheme2=T.FDM_COM6_DIRECT)
    go = O.FdmPlan.from_tables(tab, mode2=O.FDM_COM6_DIRECT)
    rng = np.random.default_rng(3)
    u, v = rng.uniform(-1, 1, nx * ny * nz), rng.uniform(-1, 1, nx * ny * nz)
    du, dv = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    res, tmp = torch.empty_like(du), torch.empty_like(du)
    T.OPR_Partial_X(T.OPR_P2, nx, ny, nz, 0, gp, du, res, tmp)
    assert load().tlab_last_kernel_path() == 1
    assert rel_err(res.cpu().numpy(), O.opr_partial(1, O.OPR_P2, nx, ny, nz, 0, go, u)[0]) <= 1e-12
    T.OPR_Burgers_X(T.OPR_B_U_IN, 1e-2, nx, ny, nz, 0, gp, du, dv, res, tmp)
    assert rel_err(res.cpu().numpy(), O.opr_burgers(1, nx, ny, nz, 0, go, 1e-2, u, v)[0]) <= 1e-12
