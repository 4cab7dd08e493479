"""GPU test of the z-slab (multi-GPU) algorithm on ONE device: all npro_k ranks are simulated in this process
(LoopbackComm), every rank runs exactly the code a real rank runs (local x/y operators, K-transposes, z-operators on the
transposed layout, Poisson with its own kz range and singular-mode ownership), and the result must equal the single-domain
substep to round-off.  Only the collective itself is replaced by direct copies."""
import numpy as np
import pytest
from scatter import substep_scatter, one_ulp_noise, bound, ref_of
import cases as C

REF_HYPER = 0.1      # wall closure of the flang-built reference (defect 1); the driver classes default to the consistent 0.0

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


_ORACLE = {}


@pytest.mark.parametrize("P,nx,ny,nz,bcs,zmode,zchunk", [
    (2, 64, 32, 32, "noslip", "auto", 0), (4, 32, 64, 64, "noslip", "auto", 0), (8, 64, 32, 64, "noslip", "auto", 0), (4, 64, 32, 32, "freeslip", "auto", 0),
    # thick slabs: the transpose-free algorithm (halo planes + interface values from the neighbours, kx-pencil Poisson)
    (2, 32, 16, 128, "noslip", "halo", 0), (2, 32, 16, 128, "noslip", "halo", 16), (4, 64, 24, 256, "freeslip", "halo", 0), (8, 32, 16, 512, "noslip", "halo", 0),
    (2, 32, 16, 128, "noslip", "transpose", 0), (3, 48, 16, 192, "noslip", "halo", 0)])
def test_slab_substep_equals_single_domain(T, P, nx, ny, nz, bcs, zmode, zchunk):
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    case = C.slab(P, nx, ny, nz, bcs)            # (the inputs live in tests/cases.py: the reference-made yardstick of the case is made from the same)
    x, y, z, visc, sc = (case[k] for k in ("x", "y", "z", "visc", "sc"))
    fields = case["q0"] + case["s0"]
    one = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    slab = SlabDns(LoopbackComm(P), x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, zmode=zmode, zchunk=zchunk, hyper_bc1_ext=REF_HYPER)
    assert slab.zmode == ("transpose" if nz // P < 56 else zmode if zmode != "auto" else "halo")
    if bcs == "freeslip":
        one.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
        slab.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
    for i in range(3):
        t = torch.from_numpy(fields[i]).cuda()
        one.q[i].copy_(t); slab.scatter("q", i, t)
    t = torch.from_numpy(fields[3]).cuda()
    one.s[0].copy_(t); slab.scatter("s", 0, t)
    dtime = 2e-3
    for k in range(2):
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)
        slab.substep_of_cycle(k, dtime)
    # the bound: max(1e-12, 2 x the ORACLE's own scatter under one ulp of input noise) (tests/scatter.py); the slabs are held to it against the
    # single domain and against the oracle itself
    from oracle.tlab_oracle_rhs import DnsOracle
    from tlab_amd.dns import velocity_bcs

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        if bcs == "freeslip":
            o.flow_jmin = o.flow_jmax = velocity_bcs("freeslip"); o.scal_jmin, o.scal_jmax = [4], [3]
        return o
    key = (nx, ny, nz, bcs, P)
    if key not in _ORACLE:
        assert [(dtime * one.kdt[k], one.kco[k], True) for k in range(2)] == case["sched"]
        _ORACLE[key] = substep_scatter(make_oracle, fields[:3], fields[3:], case["sched"], nsamples=3)
    B, S = _ORACLE[key]
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s), ("hs", one.hs)):
        for i, rf in enumerate(ref):
            got = torch.cat([slab.st[r][name][i] for r in range(P)])
            tol = bound(S[1][name][i], ref=ref_of(case["key"], 1, name, i))
            err = float((got - rf).abs().max() / rf.abs().max())
            assert err <= tol, ("slab vs single domain", name, i, err, tol)
            ob = torch.from_numpy(B[1][name][i]).cuda()
            err = float((got - ob).abs().max() / ob.abs().max())
            assert err <= tol, ("slab vs oracle", name, i, err, tol)


def test_thin_slabs_refuse_halo_mode(T):
    from tlab_amd.parallel import SlabDns, LoopbackComm
    x = np.arange(32) / 32.0
    y = np.arange(16) / 15.0
    z = np.arange(64) / 64.0
    with pytest.raises(T.TlabError):
        SlabDns(LoopbackComm(4), x, y, z, zmode="halo", hyper_bc1_ext=REF_HYPER)          # kmax = 16: slab separators still couple at 1e-7
    assert SlabDns(LoopbackComm(4), x, y, z, zmode="auto", hyper_bc1_ext=REF_HYPER).zmode == "transpose"
    z48 = np.arange(192) / 192.0                                 # 48 planes per slab: 0.38^47 is still above the 1e-19 gate
    assert SlabDns(LoopbackComm(4), x, y, z48, zmode="auto", hyper_bc1_ext=REF_HYPER).zmode == "transpose"


def test_full_size_eight_slabs_equal_single_domain(T):
    """The strong-scaling case of the benchmark itself: 512^3 split into 8 slabs of 64 planes (halo mode, kx-pencils 33 + 7 x 32,
    k_zslab with two 32-row sub-chunks, k_fftz of length 512) against the single-domain driver, after one RK3 step."""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    n, P = 512, 8
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(8)
    Y = torch.arange(n, dtype=torch.float64, device="cuda").view(1, n, 1) / (n - 1)
    wall = torch.sin(np.pi * Y)
    fields = [((torch.rand(n, n, n, dtype=torch.float64, device="cuda", generator=gen) - 0.5) * wall).reshape(-1) for _ in range(4)]
    one = Dns(x, y, x.copy(), nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=REF_HYPER)
    for t, f in zip(one.q + one.s, fields):
        t.copy_(f)
    one.TIME_RUNGEKUTTA(1e-3)
    ref = [t.clone() for t in one.q + one.s]
    # conditioning at a size the oracle cannot reach: the single-domain path (itself held to the oracle's scatter at smaller sizes) re-run from
    # fields moved by one ulp of white noise
    for t, f in zip(one.q + one.s, fields):
        r = torch.randint(-1, 2, f.shape, device="cuda", generator=gen)
        t.copy_(torch.where(r > 0, torch.nextafter(f, torch.full_like(f, 1e300)), torch.where(r < 0, torch.nextafter(f, torch.full_like(f, -1e300)), f)))
    one.TIME_RUNGEKUTTA(1e-3)
    scat = [float((t - rf).abs().max() / rf.abs().max()) for t, rf in zip(one.q + one.s, ref)]
    print("512^3 one-ulp scatter of the single-domain step:", ["%.1e" % v for v in scat])
    del one
    torch.cuda.empty_cache()
    slab = SlabDns(LoopbackComm(P), x, y, x.copy(), nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=REF_HYPER)
    assert slab.zmode == "halo"
    for i in range(3):
        slab.scatter("q", i, fields[i])
    slab.scatter("s", 0, fields[3])
    for k in range(3):
        slab.substep_of_cycle(k, 1e-3)
    for i, rf in enumerate(ref):
        name, idx = ("q", i) if i < 3 else ("s", 0)
        got = torch.cat([slab.st[r][name][idx] for r in range(P)])
        err = float((got - rf).abs().max() / rf.abs().max())
        assert err <= bound(scat[i]), (name, idx, err, scat[i])


@pytest.mark.parametrize("P,nz,zmode", [(2, 128, "halo"), (4, 64, "transpose")])
def test_slab_monitors_equal_single_domain(T, P, nz, zmode):
    """TIME_COURANT (time.f90:365-548, MPI_MAX :522) and the dilatation bounds of DNS_BOUNDS_CONTROL on slabs against the single domain."""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    nx, ny = 32, 24
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz * 3.0
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    rng = np.random.default_rng(nz)
    one = Dns(x, y, z, nscal=1, visc=1.0 / 300.0, schmidt=(0.5,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    slab = SlabDns(LoopbackComm(P), x, y, z, nscal=1, visc=1.0 / 300.0, schmidt=(0.5,), yuniform=False, zmode=zmode, hyper_bc1_ext=REF_HYPER)
    assert slab.zmode == zmode
    for i in range(3):
        t = torch.from_numpy(rng.uniform(-1, 1, nx * ny * nz)).cuda()
        one.q[i].copy_(t); slab.scatter("q", i, t)
    (a1, a2), dta = one.TIME_COURANT(1.2, 0.3)
    (b1, b2), dtb = slab.TIME_COURANT(1.2, 0.3)
    assert a1 == b1 and a2 == b2 and dta == dtb            # maxima of the same numbers
    dmin, dmax = one.dilatation_bounds()
    smin, smax = slab.dilatation_bounds()
    scale = max(abs(dmin), abs(dmax))
    assert abs(dmin - smin) <= 1e-12 * scale and abs(dmax - smax) <= 1e-12 * scale


@pytest.mark.parametrize("P,nz,zmode", [(2, 128, "halo"), (4, 64, "transpose")])
def test_slabs_with_the_direct_schemes_equal_single_domain(T, P, nz, zmode):
    """The scheme set of examples/Case81-93 on slabs: SpaceOrder2 = CompactDirect6 in y (host tables, tests/golden/direct_y.npz) and
    EllipticOrder = CompactDirect6 (OPR_Poisson_FourierXZ_Direct: per-mode FDM_Int2 on the pencils / transposed slabs, ONE field on the way
    back, dp/dy = OPR_Partial_Y(p) on the slab) against the single-domain driver."""
    import os
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct_y.npz"))
    nx, ny = 32, 64
    tab = {k[len("ny%d_" % ny):]: G[k] for k in G.files if k.startswith("ny%d_" % ny)}
    x, y, z = np.arange(nx) / nx * 2.0, tab["nodes"], np.arange(nz) / nz
    mk = lambda: [T.FdmPlan(x, True, True), T.FdmPlan.from_tables(tab, False, T.FDM_COM6_JACOBIAN, T.FDM_COM6_DIRECT), T.FdmPlan(z, True, True)]   # noqa: E731
    g1, g2 = mk(), mk()
    one = Dns(x, y, z, nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, plans=g1, gy_elliptic=g1[1], hyper_bc1_ext=REF_HYPER)
    slab = SlabDns(LoopbackComm(P), x, y, z, nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, zmode=zmode, plans=g2, gy_elliptic=g2[1], hyper_bc1_ext=REF_HYPER)
    assert slab.zmode == zmode
    rng = np.random.default_rng(P)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    for i in range(4):
        a = torch.from_numpy((((np.sin(np.pi * X + i) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel())).cuda()
        (one.q[i] if i < 3 else one.s[0]).copy_(a)
        slab.scatter("q" if i < 3 else "s", i if i < 3 else 0, a)
    from oracle import tlab_oracle as O
    from oracle.tlab_oracle_rhs import DnsOracle

    def make_oracle():
        go = [O.FdmPlan(x, True, True), O.FdmPlan.from_tables(tab), O.FdmPlan(z, True, True)]
        return DnsOracle(x, y, z, nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, plans=go, gy_elliptic=go[1])
    B, S = substep_scatter(make_oracle, [t.cpu().numpy() for t in one.q], [one.s[0].cpu().numpy()], [(2e-3 * one.kdt[k], one.kco[k], True) for k in range(2)])
    for k in range(2):
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(2e-3 * one.kdt[k], one.kco[k], True)
        slab.substep_of_cycle(k, 2e-3)
    for name, ref in (("q", one.q), ("s", one.s)):
        for i, rf in enumerate(ref):
            got = torch.cat([slab.st[r][name][i] for r in range(P)])
            err = float((got - rf).abs().max() / rf.abs().max())
            assert err <= bound(S[1][name][i]), (name, i, err, S[1][name][i])        # max(1e-12, 2 x oracle scatter)


@pytest.mark.parametrize("P,nx,nz", [(2, 32, 128), (3, 48, 192), (8, 64, 512)])
def test_two_stage_pencil_poisson_is_bit_identical(T, P, nx, nz, monkeypatch):
    """The kx-pencil Poisson solve sends every rank's kx range in two halves so that the solves of one half run under the transfers of the other
    (SlabDns._poisson_pencil_staged).  Same kernels on the same modes: p and dp/dy equal those of the one-piece exchange to the bit (uneven
    pencils at P = 3 and at nx/2+1 = 33 over 8 ranks included)."""
    import torch
    from tlab_amd.parallel import SlabDns, LoopbackComm
    ny = 16
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    rng = np.random.default_rng(P + nx)
    f = rng.uniform(-1, 1, nx * ny * nz)
    hb, ht = rng.uniform(-1, 1, nx * nz), rng.uniform(-1, 1, nx * nz)
    out = {}
    for stages in ("1", "2"):
        monkeypatch.setenv("TLAB_PENCIL_STAGES", stages)
        slab = SlabDns(LoopbackComm(P), x, y, z, nscal=1, yuniform=False, zmode="halo", hyper_bc1_ext=REF_HYPER)
        assert slab.stages == int(stages)
        kmax, n = slab.kmax, slab.n
        for r in range(P):
            S = slab.st[r]
            S["txc"][0][:n].copy_(torch.from_numpy(f[r * n:(r + 1) * n]))
            S["hb"].copy_(torch.from_numpy(hb.reshape(nz, nx)[r * kmax:(r + 1) * kmax].ravel()))
            S["ht"].copy_(torch.from_numpy(ht.reshape(nz, nx)[r * kmax:(r + 1) * kmax].ravel()))
        slab._poisson_pencil()
        torch.cuda.synchronize()
        out[stages] = [torch.cat([slab.st[r]["txc"][i][:n] for r in range(P)]).cpu().numpy() for i in (0, 2)]
    assert np.isfinite(out["2"][0]).all() and np.abs(out["2"][0]).max() > 0
    assert np.array_equal(out["1"][0], out["2"][0]) and np.array_equal(out["1"][1], out["2"][1])


@pytest.mark.parametrize("P,nz,zmode", [(2, 128, "halo"), (4, 64, "transpose")])
def test_slabs_track_the_single_domain_over_many_steps(T, P, nz, zmode):
    """20 Runge-Kutta steps (60 substeps, the time step from the slab monitors each step): nothing in the slab driver's step-to-step state (fresh
    tendencies, halo planes, interface messages, staged pencil buffers) drifts away from the single domain."""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    nx, ny = 32, 24
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    rng = np.random.default_rng(P + 40)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    fields = [((np.sin(np.pi * X + k) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(4)]
    kw = dict(nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, hyper_bc1_ext=0.0)
    one = Dns(x, y, z, **kw)
    slab = SlabDns(LoopbackComm(P), x, y, z, zmode=zmode, **kw)
    for i in range(3):
        t = torch.from_numpy(fields[i]).cuda()
        one.q[i].copy_(t); slab.scatter("q", i, t)
    t = torch.from_numpy(fields[3]).cuda()
    one.s[0].copy_(t); slab.scatter("s", 0, t)
    # a second single domain started one ulp of white noise away: how far 60 substeps of this flow carry a last-bit difference is measured,
    # not assumed (the trajectories separate at the flow's own rate; tests/scatter.py for the one-substep version on the oracle)
    two = Dns(x, y, z, **kw)
    rng2 = np.random.default_rng(99)
    for i in range(4):
        (two.q[i] if i < 3 else two.s[0]).copy_(torch.from_numpy(one_ulp_noise(fields[i], rng2)).cuda())
    dts = []
    for step in range(20):
        _, dt = slab.TIME_COURANT(1.2, 0.25)
        _, dt1 = one.TIME_COURANT(1.2, 0.25)
        assert abs(dt - dt1) <= 1e-9 * dt1
        one.TIME_RUNGEKUTTA(dt)
        two.TIME_RUNGEKUTTA(dt)
        for k in range(3):
            slab.substep_of_cycle(3 * step + k, dt)
    for name, ref, alt in (("q", one.q, two.q), ("s", one.s, two.s)):
        for i, rf in enumerate(ref):
            got = torch.cat([slab.st[r][name][i] for r in range(P)])
            assert bool(torch.isfinite(got).all())
            scat = float((alt[i] - rf).abs().max() / rf.abs().max())
            err = float((got - rf).abs().max() / rf.abs().max())
            assert err <= bound(scat), (name, i, err, scat)
