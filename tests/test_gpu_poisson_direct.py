"""GPU parity of the DIRECT elliptic solver (EllipticOrder = CompactDirect6, OPR_Poisson_FourierXZ_Direct, opr_elliptic.f90:368-455; used by
examples/Case81-93) through the C ABI against the oracle, whose per-mode arithmetic (FDM_Int2_*) is pinned bit-exact to the reference build.
The y plan tables of the direct scheme come from the reference-generated fixture tests/golden/direct_y.npz."""
import numpy as np
import pytest
from conftest import golden_files, rel_err
from oracle import tlab_oracle as O
from oracle import tlab_oracle_poisson as OP

pytestmark = pytest.mark.gpu
TOL = 1e-13     # p with the MARCHING kernel (k_int2): it repeats the reference's operations bit for bit; what is left is rocFFT against numpy.fft (~1e-15)


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


def _tab(ny):
    if ny == 40:      # EllipticOrder = CompactDirect4: the (CompactJacobian6, CompactDirect4) y plan of tests/golden/poisson_direct_modes_c4_40.npz
        g = np.load([f for f in golden_files("poisson_direct_modes_c4")][0])
        tab = {k[len("tab_"):]: g[k] for k in g.files if k.startswith("tab_")}
        tab["nodes"] = g["y"]
        return tab, 17
    g = np.load(golden_files("direct_y")[0])
    return {k[len("ny%d_" % ny):]: g[k] for k in g.files if k.startswith("ny%d_" % ny)}, 16


def _setup(T, nx, ny, nz, seed):
    tab, mode2 = _tab(ny)
    y = tab["nodes"]
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * np.pi if nz > 1 else np.zeros(1)
    ogx, ogz = O.FdmPlan(x, True, True), O.FdmPlan(z, True, True) if nz > 1 else None
    ogy = O.FdmPlan.from_tables(tab, mode2=mode2)
    gx = T.FdmPlan(x, True, True)
    gz = T.FdmPlan(z, True, True) if nz > 1 else T.FdmPlan(np.zeros(1), True, True)
    gy = T.FdmPlan.from_tables(tab, scheme1=6, scheme2=mode2)        # the derivative plan of the run: (CompactJacobian6, CompactDirect6 | 4)
    rng = np.random.default_rng(seed)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    f = (np.sin(X) * np.cos(2 * Z) * np.exp(Y) + 0.3 * rng.uniform(-1, 1, X.shape)).ravel()
    hb, ht = rng.uniform(-1, 1, (nz, nx)), rng.uniform(-1, 1, (nz, nx))
    return (ogx, ogy, ogz), (gx, gy, gz), f, hb, ht


@pytest.mark.parametrize("nx,ny,nz,ibc", [(16, 24, 8, 3), (16, 24, 8, 0), (16, 24, 8, 1), (16, 24, 8, 2), (32, 64, 16, 3), (16, 64, 1, 3),
                                          (32, 128, 8, 3), (32, 128, 8, 0), (64, 512, 16, 3),
                                          (16, 40, 8, 3), (16, 40, 8, 0), (16, 40, 8, 1), (16, 40, 8, 2)])      # 40: CompactDirect4
def test_poisson_direct_matches_oracle(T, nx, ny, nz, ibc):
    import torch
    (ogx, ogy, ogz), (gx, gy, gz), f, hb, ht = _setup(T, nx, ny, nz, ny + ibc)
    oplan = OP.PoissonDirectPlan(ogx, ogy, ogz if nz > 1 else ogx, nx, ny, nz)
    from scatter import scatter_of, bound, ref_of
    (p_ref, dp_ref), (sc_p, sc_dp) = scatter_of(lambda f_, hb_, ht_: OP.opr_poisson_fxz_direct(oplan, f_, hb_, ht_, ibc), [f, hb, ht], nsamples=2)
    plan = T.PoissonPlan(gx, gy, gz, nx, ny, nz, gy_elliptic=gy)
    assert plan.direct
    dev = "cuda"
    p = torch.from_numpy(f.copy()).to(dev)
    tmp1 = torch.zeros(plan.isize_txc_field, dtype=torch.float64, device=dev)
    tmp2 = torch.zeros_like(tmp1)
    dpdy = torch.zeros_like(p)
    T.OPR_Poisson(plan, nx, ny, nz, ibc, p, tmp1, tmp2, torch.from_numpy(hb.ravel().copy()).to(dev), torch.from_numpy(ht.ravel().copy()).to(dev), dpdy)
    torch.cuda.synchronize()
    assert rel_err(p.cpu().numpy(), p_ref) <= TOL, rel_err(p.cpu().numpy(), p_ref)
    # dp/dy = OPR_Partial_Y(p): differentiating the 1e-15 FFT noise of p on the stretched grid (h_min ~ 1/(4 ny)) costs ~ny digits -- on the oracle
    # too: the bound is max(1e-12, 2 x the oracle's own scatter under one ulp of input noise) (tests/scatter.py)
    # (the reference's own OPR_Partial_Y on the same pressure, two builds / one ulp of noise: tests/golden/yardsticks.json, recorded next to the error)
    assert rel_err(dpdy.cpu().numpy(), dp_ref) <= bound(sc_dp, ref=ref_of("poisson_direct.dpdy[%d-%d-%d-%d]" % (nx, ny, nz, ibc), 0, "dpdy", 0)), (rel_err(dpdy.cpu().numpy(), dp_ref), sc_dp)


@pytest.mark.parametrize("ibc", [0, 1, 2, 3])
def test_marching_and_chunked_direct_solvers_agree(T, ibc):
    """k_int2c (8-row chunks, parallel scan) is the default where ny is a multiple of 8; TLAB_INT2_CHUNKED=0 (read per call) selects the marching
    k_int2, which repeats the reference's operation order.  Same plan, both routes: they differ by the association of the substitution sums only."""
    import os
    import torch
    nx, ny, nz = 32, 128, 8
    (ogx, ogy, ogz), (gx, gy, gz), f, hb, ht = _setup(T, nx, ny, nz, 40 + ibc)
    plan = T.PoissonPlan(gx, gy, gz, nx, ny, nz, gy_elliptic=gy)
    tmp1 = torch.zeros(plan.isize_txc_field, dtype=torch.float64, device="cuda"); tmp2 = torch.zeros_like(tmp1)
    out = {}
    try:
        for route in ("1", "0"):
            os.environ["TLAB_INT2_CHUNKED"] = route
            p = torch.from_numpy(f.copy()).cuda(); dpdy = torch.zeros_like(p)
            T.OPR_Poisson(plan, nx, ny, nz, ibc, p, tmp1, tmp2, torch.from_numpy(hb.ravel().copy()).cuda(), torch.from_numpy(ht.ravel().copy()).cuda(), dpdy)
            out[route] = p.cpu().numpy()
    finally:
        os.environ.pop("TLAB_INT2_CHUNKED", None)
    oplan = OP.PoissonDirectPlan(ogx, ogy, ogz, nx, ny, nz)
    p_ref, _ = OP.opr_poisson_fxz_direct(oplan, f, hb, ht, ibc)
    assert rel_err(out["0"], p_ref) <= TOL and rel_err(out["1"], p_ref) <= 1e-12
    assert not np.array_equal(out["0"], out["1"])         # (two different kernels did run)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,nz,ibc,alpha", [(16, 24, 8, 0, -12.5), (16, 64, 8, 3, -400.0), (32, 128, 8, 1, -3.0), (16, 64, 1, 2, -50.0), (64, 512, 16, 0, -1.0e4)])
def test_helmholtz_direct_matches_oracle(T, nx, ny, nz, ibc, alpha):
    import torch
    (ogx, ogy, ogz), (gx, gy, gz), f, hb, ht = _setup(T, nx, ny, nz, ny + ibc + 7)
    oplan = OP.PoissonDirectPlan(ogx, ogy, ogz if nz > 1 else ogx, nx, ny, nz)
    a_ref = OP.opr_helmholtz_fxz_direct(oplan, f, hb, ht, ibc, alpha)
    plan = T.PoissonPlan(gx, gy, gz, nx, ny, nz, gy_elliptic=gy)
    a = torch.from_numpy(f.copy()).cuda()
    tmp1 = torch.zeros(plan.isize_txc_field, dtype=torch.float64, device="cuda")
    tmp2 = torch.zeros_like(tmp1)
    T.OPR_Helmholtz(plan, nx, ny, nz, ibc, alpha, a, tmp1, tmp2, torch.from_numpy(hb.ravel().copy()).cuda(), torch.from_numpy(ht.ravel().copy()).cuda())
    torch.cuda.synchronize()
    assert rel_err(a.cpu().numpy(), a_ref) <= TOL, rel_err(a.cpu().numpy(), a_ref)


@pytest.mark.gpu
def test_factorized_helmholtz_refuses_mixed_boundary_types(T):
    """OPR_Helmholtz_FourierXZ_Factorize knows BCS_NN and BCS_DD (opr_elliptic.f90:524-532)."""
    import torch
    x = np.arange(16) / 16.0
    y = np.arange(24) / 23.0
    gx, gy = T.FdmPlan(x, True, True), T.FdmPlan(y, False, True)
    plan = T.PoissonPlan(gx, gy, gx, 16, 24, 16)
    a = torch.zeros(16 * 24 * 16, dtype=torch.float64, device="cuda")
    t1 = torch.zeros(plan.isize_txc_field, dtype=torch.float64, device="cuda"); t2 = torch.zeros_like(t1)
    hb = torch.zeros(256, dtype=torch.float64, device="cuda")
    with pytest.raises(T.TlabError):
        T.OPR_Helmholtz(plan, 16, 24, 16, 1, -1.0, a, t1, t2, hb, hb.clone())
    with pytest.raises(T.TlabError):      # lambda - alpha must be positive (the mean mode has lambda = 0)
        T.OPR_Helmholtz(plan, 16, 24, 16, 3, 0.0, a, t1, t2, hb, hb.clone())


def test_direct_plan_needs_the_direct_tables(T):
    x = np.arange(16) / 16.0
    y = np.arange(24) / 23.0
    gx, gy = T.FdmPlan(x, True, True), T.FdmPlan(y, False, True)
    with pytest.raises(T.TlabError):
        T.PoissonPlan(gx, gy, gx, 16, 24, 16, gy_elliptic=gy)       # Jacobian second derivative: not a CompactDirect plan


def test_full_size_direct_solver_satisfies_the_discrete_equation(T):
    """512 x 512 x 64 (the y line of the benchmark box): the interior rows of (d2/dx2 + d2/dy2 + d2/dz2) p reproduce the forcing with the
    device's own second-derivative operators (direct scheme in y), a size-independent property."""
    import torch
    nx, ny, nz = 512, 512, 64
    (_, _, _), (gx, gy, gz), _, _, _ = _setup(T, 16, ny, 8, 0)
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * np.pi
    gx, gz = T.FdmPlan(x, True, True), T.FdmPlan(z, True, True)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    n = nx * ny * nz
    f = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    f3 = f.view(nz, ny, nx)
    f3 -= f3.mean(dim=(0, 2), keepdim=True)                         # the mean mode is pinned by p = 0 at the bottom, not by its Neumann datum
    hb = torch.zeros(nx * nz, dtype=torch.float64, device="cuda")
    ht = torch.zeros_like(hb)
    plan = T.PoissonPlan(gx, gy, gz, nx, ny, nz, gy_elliptic=gy)
    p = f.clone()
    tmp1 = torch.zeros(plan.isize_txc_field, dtype=torch.float64, device="cuda")
    tmp2 = torch.zeros_like(tmp1)
    T.OPR_Poisson(plan, nx, ny, nz, 3, p, tmp1, tmp2, hb, ht, None)
    lap = torch.zeros_like(p)
    r = torch.zeros_like(p)
    t = torch.zeros_like(p)
    for d, g, part in ((1, gx, T.OPR_Partial_X), (2, gy, T.OPR_Partial_Y), (3, gz, T.OPR_Partial_Z)):
        part(T.OPR_P2, nx, ny, nz, 0, g, p, r, t)
        lap += r
    a, b = lap.view(nz, ny, nx)[:, 3:ny - 3], f.view(nz, ny, nx)[:, 3:ny - 3]
    err = float((a - b).abs().max() / b.abs().max())
    assert err <= 1e-7, err
