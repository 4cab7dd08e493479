"""The reference's own validation programs, run against the device operators through the Python mirror of the operator interface:

  src/valid/fdm/vpartial.f90:98-175     Gaussian profile, first and second derivative against the analytic ones (the program prints the error
                                        and has no threshold; the thresholds here are the truncation errors the oracle itself shows)
  src/valid/burgers/vburgers.f90:73-160 OPR_Burgers_{X,Y,Z}(OPR_B_SELF) against visc * d2 - a * d1 assembled from OPR_Partial(OPR_P2_P1),
                                        relative L2 error per direction (expected ~1e-15)
  (src/valid/elliptic/vpoisson.f90 -> tests/test_gpu_poisson.py; operators/opr_check.f90 -> tests/test_slab_transposes_cpu.py)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


def _grids(nx, ny, nz):
    x = np.arange(nx) / nx
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(2))      # the stretched grid of vpartial's nonuniform case
    return x, y, z


@pytest.mark.parametrize("n", [128, 512])
def test_vpartial_gaussian(T, n):
    """vpartial.f90:100-111: u = exp(-(x - x0 L)^2 / (2 (L/wk)^2)), x0 = 0.75, wk = 1, along the non-periodic stretched direction: first
    derivative on every row.  (The second derivative of the default CompactJacobian6Hyper scheme is NOT checked against the analytic one in
    a non-periodic direction: its wall rows carry the reference's out-of-bounds read, fdm_com2_jacobian.f90:224, DESIGN.md section 2, and
    the implicit solve spreads that inwards -- the oracle, i.e. the reference's own arithmetic, is off by O(1e3) ten rows from the wall.
    It is checked in the periodic direction below, with vpartial's single-mode function.)"""
    import torch
    from oracle import tlab_oracle as O
    nx, nz = 64, 8
    x, y, z = _grids(nx, n, nz)
    g = T.FdmPlan(y, False, False)
    og = O.FdmPlan(y, False, False)
    L = y[-1] - y[0]
    x0, wk = 0.75, 1.0
    u1 = np.exp(-(y - x0 * L) ** 2 / (2.0 * (L / wk) ** 2))
    d1 = -(y - x0 * L) / (L / wk) ** 2 * u1
    rep = lambda a: np.broadcast_to(a[None, :, None], (nz, n, nx)).ravel().copy()      # noqa: E731
    u = torch.from_numpy(rep(u1)).cuda()
    r = torch.zeros_like(u); t = torch.zeros_like(u)
    T.OPR_Partial_Y(T.OPR_P1, nx, n, nz, 0, g, u, r, t)
    e1 = float(np.abs(r.cpu().numpy() - rep(d1)).max() / np.abs(d1).max())
    o1 = float(np.abs(O.opr_partial(2, 1, nx, n, nz, 0, og, rep(u1))[0] - rep(d1)).max() / np.abs(d1).max())     # the reference's own arithmetic
    assert e1 <= 1.01 * o1 + 1e-13, (e1, o1)
    assert e1 <= (1e-5 if n == 128 else 2e-7), e1            # 3rd-order wall closures on the coarse end of the stretched grid


@pytest.mark.parametrize("n", [128, 512])
def test_vpartial_single_mode_periodic(T, n):
    """vpartial.f90:101-104 (the single-mode case): u = 1 + sin(2 pi wk x / L) in the periodic direction, both derivatives (OPR_P2_P1)."""
    import torch
    ny, nz, wk = 16, 8, 3.0
    x = np.arange(n) / n * 2.0
    L = 2.0
    g = T.FdmPlan(x, True, True)
    u1 = 1.0 + np.sin(2 * np.pi / L * wk * x)
    d1 = (2 * np.pi / L * wk) * np.cos(2 * np.pi / L * wk * x)
    d2 = -(2 * np.pi / L * wk) ** 2 * np.sin(2 * np.pi / L * wk * x)
    rep = lambda a: np.broadcast_to(a[None, None, :], (nz, ny, n)).ravel().copy()      # noqa: E731
    u = torch.from_numpy(rep(u1)).cuda()
    r = torch.zeros_like(u); t = torch.zeros_like(u)
    T.OPR_Partial_X(T.OPR_P2_P1, n, ny, nz, 0, g, u, r, t)
    e1 = float(np.abs(t.cpu().numpy() - rep(d1)).max() / np.abs(d1).max())
    e2 = float(np.abs(r.cpu().numpy() - rep(d2)).max() / np.abs(d2).max())
    h = 2 * np.pi * wk / n                                   # radians per grid point
    assert e1 <= 2.0 * h ** 6 + 1e-13 and e2 <= 2.0 * h ** 6 + 1e-12, (e1, e2, h ** 6)       # sixth-order schemes


@pytest.mark.parametrize("nx,ny,nz", [(128, 96, 64), (512, 512, 16)])
def test_vburgers(T, nx, ny, nz):
    """vburgers.f90:76-160: relative L2 error of OPR_Burgers (SELF) against visc * P2 - a * P1 from OPR_Partial(OPR_P2_P1), per direction."""
    import torch
    x, y, z = _grids(nx, ny, nz)
    g = [T.FdmPlan(x, True, True), T.FdmPlan(y, False, False), T.FdmPlan(z, True, True)]
    rng = np.random.default_rng(nx)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    a = torch.from_numpy((np.sin(2 * np.pi * X) * np.cos(4 * np.pi * Y) * np.sin(6 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()).cuda()
    visc = 1.0 / 5000.0
    b = torch.zeros_like(a); c = torch.zeros_like(a); tmp = torch.zeros_like(a)
    part = (T.OPR_Partial_X, T.OPR_Partial_Y, T.OPR_Partial_Z)
    burg = (T.OPR_Burgers_X, T.OPR_Burgers_Y, T.OPR_Burgers_Z)
    for d in range(3):
        part[d](T.OPR_P2_P1, nx, ny, nz, 0, g[d], a, b, c)               # b = d2, c = d1
        direct = b * visc - a * c
        burg[d](T.OPR_B_SELF, visc, nx, ny, nz, 0, g[d], a, a, c, tmp)
        err = float(torch.sqrt(((c - direct) ** 2).sum()) / torch.sqrt((direct ** 2).sum()))
        assert err <= 1e-13, ("XYZ"[d], err)
