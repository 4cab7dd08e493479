"""GPU parity test of the full RK substep (RHS_GLOBAL_INCOMPRESSIBLE_1 + update) against the numpy oracle composition,
and of the invariant the projection must satisfy (interior divergence of the new velocity ~ round-off)."""
import numpy as np
import pytest
from conftest import rel_err
from scatter import substep_scatter, bound, FLOOR, cpu_port_factory, CPU_PORT_SOURCE, ref_of
import cases as C
from cases import grids, init_fields      # noqa: F401  (the inputs of the composed-path cases live in tests/cases.py: the yardstick generator uses them too)

REF_HYPER = 0.1      # wall closure of the flang-built reference (defect 1); the driver classes default to the consistent 0.0


def check_state(d, B, S, k, names=("q", "hq", "s", "hs"), tag="", factor=2.0, source=None, key=None):
    """Device state after substep k against the oracle's, each field within max(1e-12, factor x the oracle's own one-ulp scatter) (tests/scatter.py).
    key: the case of tests/cases.py -- the difference between two builds of the reference on it is recorded next to the error (tests/golden/yardsticks.json)."""
    for name in names:
        for i, (b, sc) in enumerate(zip(B[k][name], S[k][name])):
            e = rel_err(getattr(d, name)[i].cpu().numpy(), b)
            bd = bound(sc, factor, source or "oracle one-ulp scatter", ref=ref_of(key, k, name, i))
            assert e <= bd, (tag, k, name, i, "err %.2e" % e, "oracle scatter %.2e" % sc)


BIG = 4_000_000      # points from which the C + OpenMP port (oracle/tlab_cpu.c, 16 threads on the GPU box) is the checker instead of the numpy oracle


def oracle_factory(x, y, z, nscal, visc, sc, stretch):
    """(make_oracle, yardstick name): no-slip walls, default schemes.  The numpy oracle (single-threaded) below BIG points; above, the C / OpenMP
    restatement, which the CPU suite holds to the reference's golden vectors and to the numpy oracle (tests/test_cpu_baseline.py)."""
    from oracle.tlab_oracle_rhs import DnsOracle
    if len(x) * len(y) * len(z) >= BIG:
        return cpu_port_factory(x, y, z, nscal, visc, sc, not stretch), CPU_PORT_SOURCE
    return (lambda: DnsOracle(x, y, z, nscal=nscal, visc=visc, schmidt=sc, yuniform=not stretch)), None

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


_ORACLE_CACHE = {}


def oracle_substeps(key, make_oracle, q0, s0, schedule, nsamples=1):
    """substep_scatter, cached per test shape (the fuse / exact parametrizations share one oracle run)."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = substep_scatter(make_oracle, q0, s0, schedule, nsamples)
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("hyper", [REF_HYPER, 0.0])      # 0.0: the consistent closure, which bench.py times
@pytest.mark.parametrize("nx,ny,nz,stretch", [(32, 40, 16, True), (64, 32, 32, False), (256, 64, 64, True)])
def test_substep_vs_oracle(T, nx, ny, nz, stretch, hyper, fuse):
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    case = C.rhs_substep(nx, ny, nz, stretch, hyper=None if hyper == REF_HYPER else hyper)
    x, y, z, visc, sc, q0, s0, sched = (case[k] for k in ("x", "y", "z", "visc", "sc", "q0", "s0", "sched"))
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=hyper)
    d.set_fusion(fuse)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    assert [s_[0] for s_ in sched] == [2e-3 * d.kdt[k] for k in range(2)] and [s_[1] for s_ in sched] == [d.kco[k] for k in range(2)]
    # two RK3 substeps incl. the tendency scaling in between (time.f90:220-298)
    B, S = oracle_substeps(("sub", nx, ny, nz, stretch, hyper), lambda: DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=hyper),
                           q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, key=case["key"])      # (the reference-made figure is that of the SAME closure: for 0.0 the reference's routines on its own plan with the one out-of-bounds entry replaced, tests/cases.py)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("vel,scal", [(("freeslip", "freeslip"), ("neumann", "dirichlet")), (("noslip", "freeslip"), ("dirichlet", "neumann"))])
def test_substep_with_neumann_walls_vs_oracle(T, vel, scal, fuse):
    """Free-slip walls / Neumann scalars (the reference's default VelocityJmin = freeslip, boundary_bcs.f90:114):
    BOUNDARY_BCS_NEUMANN_Y sets the wall tendencies (rhs_global_incompressible_1.f90:363-396)."""
    import torch
    from tlab_amd.dns import Dns, velocity_bcs, scalar_bcs
    from oracle.tlab_oracle_rhs import DnsOracle
    case = C.rhs_neumann(vel, scal)
    nx, ny, nz, stretch = 64, 64, 32, True
    x, y, z, visc, sc, q0, s0 = (case[k] for k in ("x", "y", "z", "visc", "sc", "q0", "s0"))
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=REF_HYPER)
    d.set_fusion(fuse)
    d.set_bcs(vel[0], vel[1], scal[0], scal[1])

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch)
        o.flow_jmin, o.flow_jmax = velocity_bcs(vel[0]), velocity_bcs(vel[1])
        o.scal_jmin, o.scal_jmax = [scalar_bcs(scal[0])], [scalar_bcs(scal[1])]
        return o
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    dtime = 2e-3
    sched = [(dtime * d.kdt[k], d.kco[k], True) for k in range(2)]
    assert sched == case["sched"]
    B, S = oracle_substeps(("neumann", vel, scal), make_oracle, q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, key=case["key"])
    # the wall tendencies are not zero where Neumann was asked for, and the RHS entry point agrees with the fused substep
    hq0 = B[1]["hq"][0].reshape(nz, ny, nx)
    assert np.abs(hq0[:, -1, :]).max() > 0
    with pytest.raises(T.TlabError):
        d.set_bcs(velocity_jmin="noslip", velocity_jmax="noslip", scalar_jmin="robin")


@pytest.mark.parametrize("vel,scal", [(("freeslip", "freeslip"), ("neumann", "neumann")), (("freeslip", "noslip"), ("dirichlet", "neumann"))])
def test_neumann_wall_planes_route(T, vel, scal, monkeypatch):
    """The fused tail with free-slip walls on the fast kernels (nx = 256): the wall tendencies of u and w come from weighted sums over the rows next
    to the walls -- BOUNDARY_BCS_NEUMANN_Y's wall value as a linear functional of the line, its weights read off the library's own routine, the
    sums of the pressure differentiated as planes -- and the gradient kernels finish the fields in one pass (default); TLAB_NEUMANN_PLANES=0 keeps the
    derivative pass over each field (k_rtile<P1+neumann final>).  The two routes agree to round-off, and both with the oracle within its scatter."""
    import ctypes
    import torch
    from tlab_amd.dns import Dns, velocity_bcs, scalar_bcs
    from tlab_amd.lib import load
    from oracle.tlab_oracle_rhs import DnsOracle
    case = C.rhs_neumann(vel, scal, nx=256, seed=21)
    nx, ny, nz, stretch = 256, 64, 64, True
    x, y, z, visc, sc, q0, s0 = (case[k] for k in ("x", "y", "z", "visc", "sc", "q0", "s0"))
    L = load()
    out = {}
    for planes in ("1", "0"):
        monkeypatch.setenv("TLAB_NEUMANN_PLANES", planes)
        d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=REF_HYPER)
        d.set_bcs(vel[0], vel[1], scal[0], scal[1])
        for i in range(3):
            d.q[i].copy_(torch.from_numpy(q0[i]))
        d.s[0].copy_(torch.from_numpy(s0[0]))
        dtime = 2e-3
        sched = [(dtime * d.kdt[k], d.kco[k], True) for k in range(2)]
        L.tlab_profile_reset(); L.tlab_profile_enable(1)
        for dte, kco, scale in sched:
            d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        torch.cuda.synchronize()
        L.tlab_profile_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        L.tlab_profile_report(buf, len(buf))
        rep = buf.value.decode()
        assert ("k_wall_weighted" in rep) == (planes == "1"), rep              # the route taken
        assert ("k_rtile<P1+neumann final>" in rep) == (planes == "0"), rep   # ... for the Neumann scalar as well (its interior rides on the x Burgers launch)
        out[planes] = [t.cpu().numpy().copy() for t in d.q + d.s + d.hq + d.hs]
        if planes == "1":
            def make_oracle():
                o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch)
                o.flow_jmin, o.flow_jmax = velocity_bcs(vel[0]), velocity_bcs(vel[1])
                o.scal_jmin, o.scal_jmax = [scalar_bcs(scal[0])], [scalar_bcs(scal[1])]
                return o
            assert sched == case["sched"]
            B, S = oracle_substeps(("neumann planes", vel, scal), make_oracle, q0, s0, sched, nsamples=2)
            check_state(d, B, S, 1, key=case["key"])
    for a, b in zip(out["1"], out["0"]):
        assert rel_err(a, b) <= 1e-12      # (two substeps; the sums and the sweeps round differently: measured 1.6e-13 on the tendencies)
    hq0 = out["1"][4].reshape(nz, ny, nx)
    assert np.abs(hq0[:, 0, :]).max() > 0          # a Neumann wall at jmin for u: its wall tendency is not zero


def test_rhs_entry_with_neumann_walls_vs_oracle(T):
    import torch
    from tlab_amd.dns import Dns, velocity_bcs
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 32, 48, 16
    x, y, z = grids(nx, ny, nz, False)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 9)
    d = Dns(x, y, z, nscal=1, visc=1e-3, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=1, visc=1e-3, schmidt=(1.0,), yuniform=True)
    d.set_bcs("freeslip", "noslip", "neumann", "neumann")
    o.flow_jmin, o.scal_jmin, o.scal_jmax = velocity_bcs("freeslip"), [4], [4]
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i])); o.q[i] = q0[i].copy()
    d.s[0].copy_(torch.from_numpy(s0[0])); o.s[0] = s0[0].copy()
    d.RHS_GLOBAL_INCOMPRESSIBLE_1(1e-3)
    o.rhs_global_incompressible_1(1e-3)
    # the oracle's own one-ulp scatter of the tendencies (tests/scatter.py) bounds the comparison
    from scatter import scatter_of

    def rhs(*f):
        o2 = DnsOracle(x, y, z, nscal=1, visc=1e-3, schmidt=(1.0,), yuniform=True)
        o2.flow_jmin, o2.scal_jmin, o2.scal_jmax = velocity_bcs("freeslip"), [4], [4]
        o2.q = [a.copy() for a in f[:3]]; o2.s = [f[3].copy()]
        o2.rhs_global_incompressible_1(1e-3)
        return tuple(o2.hq + o2.hs)
    base, sc = scatter_of(rhs, q0 + s0, nsamples=2)
    for i in range(3):
        assert np.array_equal(base[i], o.hq[i])
        assert rel_err(d.hq[i].cpu().numpy(), o.hq[i]) <= bound(sc[i]), (i, sc[i])
    assert rel_err(d.hs[0].cpu().numpy(), o.hs[0]) <= bound(sc[3]), sc[3]


def test_case01_shaped_two_dimensional_step(T):
    """BASELINE configs[0] = examples/Case01: 512 x 256 x 1, RungeKuttaExplicit4, free-slip walls, Neumann scalar, time step from
    TimeCFL = 1.2 (tlab.ini).  Its initial fields come from the reference's initialisation tools (out of scope), so the same
    plumbing runs on synthetic shear-layer fields: one full RK4 step + TIME_COURANT + dilatation bounds against the oracle."""
    import torch
    from tlab_amd.dns import Dns, RKM_EXP4, velocity_bcs
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 512, 256, 1
    x = np.arange(nx) / nx * 2.0
    y = np.arange(ny) / (ny - 1.0)
    z = np.zeros(1)
    rng = np.random.default_rng(101)
    Y, X = np.meshgrid(y, x, indexing="ij")
    u0 = 0.5 * np.tanh((Y - 0.5) / (2 * 0.005859375 * 8)) + 0.02 * rng.uniform(-1, 1, X.shape) * np.exp(-((Y - 0.5) / 0.1) ** 2)
    v0 = 0.02 * rng.uniform(-1, 1, X.shape) * np.exp(-((Y - 0.5) / 0.1) ** 2) * np.sin(np.pi * Y)
    s0 = 0.5 - 0.5 * np.tanh((Y - 0.5) / (2 * 0.005859375 * 8))
    d = Dns(x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP4, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True)
    d.set_bcs("freeslip", "freeslip", "neumann", "neumann")
    o.flow_jmin = o.flow_jmax = velocity_bcs("freeslip"); o.scal_jmin = o.scal_jmax = [4]
    for i, a in enumerate((u0, v0, np.zeros_like(u0))):
        d.q[i].copy_(torch.from_numpy(a.ravel())); o.q[i] = a.ravel().copy()
    d.s[0].copy_(torch.from_numpy(s0.ravel())); o.s[0] = s0.ravel().copy()
    (p1, p2), dt = d.TIME_COURANT(1.2, 0.25)
    (r1, r2), rdt = o.time_courant(1.2, 0.25)
    assert abs(dt - rdt) <= 1e-14 * rdt
    kdt, kco = d.kdt, d.kco
    assert len(kdt) == 5
    for k in range(5):                      # TIME_RUNGEKUTTA, time.f90:212-298
        last = k == 4
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dt * kdt[k], 1.0 if last else kco[k], not last)
        o.time_substep(rdt * kdt[k], 1.0 if last else kco[k], not last)
    for i in range(2):
        assert rel_err(d.q[i].cpu().numpy(), o.q[i]) <= 1e-12, i
    assert float(d.q[2].abs().max()) == 0.0 and np.abs(o.q[2]).max() == 0.0         # no z-dynamics in 2-D
    assert rel_err(d.s[0].cpu().numpy(), o.s[0]) <= 1e-12
    dmin, dmax = d.dilatation_bounds()
    ref = -o.fi_invariant_p()
    assert abs(dmin - ref.min()) <= 1e-11 * max(1.0, np.abs(ref).max()) and abs(dmax - ref.max()) <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nscal", [0, 3])
def test_other_scalar_counts(T, nscal):
    """configs[4] carries 3 active scalars (two Burgers launches per direction: 4 + 2 fields); nscal = 0 is the pure flow case."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 256, 64, 32
    x, y, z = grids(nx, ny, nz, True)
    sc = (0.7, 1.0, 2.5)[:nscal]
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 17)
    d = Dns(x, y, z, nscal=nscal, visc=1.0 / 900.0, schmidt=sc if nscal else (1.0,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=nscal, visc=1.0 / 900.0, schmidt=sc if nscal else (1.0,), yuniform=False)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i])); o.q[i] = q0[i].copy()
    for i in range(nscal):
        a = s0[0] * (1.0 + 0.3 * i) + 0.1 * i
        d.s[i].copy_(torch.from_numpy(a)); o.s[i] = a.copy()
    sched = [(2e-3 * d.kdt[k], d.kco[k], True) for k in range(2)]
    B, S = substep_scatter(lambda: DnsOracle(x, y, z, nscal=nscal, visc=1.0 / 900.0, schmidt=sc if nscal else (1.0,), yuniform=False),
                           [o.q[i] for i in range(3)], [o.s[i] for i in range(nscal)], sched, nsamples=1)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
    check_state(d, B, S, 1)


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("nx,ny,nz,nscal,stretch", [
    (1024, 512, 16, 1, False),      # configs[3]: x lines of 1024, y lines of 512 (z lines of 1024: test_gpu_slab.py)
    (2048, 1024, 8, 3, True),       # configs[4]: x lines of 2048, stretched y lines of 1024, 3 scalars
    (16, 32, 2048, 1, False),       # z lines of 2048 on one device (own z-FFT of length 2048, 64-row tiles)
    (32, 16, 1024, 1, False)])      # z lines of 1024 on one device: the first derivatives with their fused epilogues on k_htile's 32-line tiles
def test_line_lengths_of_the_large_configs(T, nx, ny, nz, nscal, stretch, exact):
    """Full substeps against the oracle with the LINE LENGTHS of configs[3] and configs[4] (BASELINE.json) and the other extents reduced so
    that the numpy oracle finishes in seconds: every kernel selection that depends on n is exercised at its real n.
    The first substep projects a field that is not solenoidal: the pressure forcing div(q)/dte is 1e4-1e5 for a pressure of 1e2, and the
    solve amplifies rounding accordingly.  The bound is therefore MEASURED: the oracle is run a second time from fields moved by one ulp of
    white noise, and the device must stay within max(1e-12, 2 x that scatter) of the oracle (tests/scatter.py); with the exact mode of the solver
    (tlab_poisson_set_exact(1): the reference's operations one by one, transforms still rocFFT's) within 1.6 x that scatter (measured on the
    bare solve: 1.5 x, tests/test_gpu_poisson.py::test_projection_forcing_within_the_oracles_own_scatter)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    case = C.rhs_lines(nx, ny, nz, nscal, stretch)
    x, y, z, visc, sc, q0, ss = (case[k] for k in ("x", "y", "z", "visc", "sc", "q0", "s0"))
    T.poisson_set_exact(exact)
    try:
        d = Dns(x, y, z, nscal=nscal, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=REF_HYPER)
    finally:
        T.poisson_set_exact(False)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    for i in range(nscal):
        d.s[i].copy_(torch.from_numpy(ss[i]))
    dt = 1e-3
    sched = [(dt * d.kdt[k], d.kco[k], True) for k in range(2)]
    assert sched == case["sched"]
    make, source = oracle_factory(x, y, z, nscal, visc, sc, stretch)
    B, S = oracle_substeps(("lines", nx, ny, nz), make, q0, ss, sched, nsamples=3 if source is None else 2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
    errs = [rel_err(d.q[i].cpu().numpy(), B[1]["q"][i]) for i in range(3)]
    print("exact" if exact else "fast ", (nx, ny, nz), "err", ["%.1e" % e for e in errs], "oracle one-ulp scatter", ["%.1e" % e for e in S[1]["q"]])
    check_state(d, B, S, 1, names=("q", "s"), tag="exact" if exact else "fast", factor=1.6 if exact else 2.0, source=source, key=case["key"])


@pytest.mark.parametrize("nx,ny,nz,nscal,stretch", [
    (64, 512, 16, 1, False),        # y lines of the headline benchmark
    (1024, 512, 16, 1, False),      # configs[3]
    (2048, 1024, 8, 3, True),       # configs[4]
    (16, 32, 2048, 1, False)])      # z lines of 2048
def test_substeps_from_an_already_projected_field(T, nx, ny, nz, nscal, stretch):
    """The state a DNS is in for all but its first substep: the oracle integrates one full Runge-Kutta step from the
    non-solenoidal test field, its state (three times projected) is handed to the device, and substeps 4-6 are compared one by one -- with the
    line lengths of the BASELINE configs.  Asserted like every composed path (max(1e-12, 2 x the oracle's own one-ulp scatter)); whether the
    north-star's 1e-12 itself holds from a projected state is RECORDED per substep (parity_table.json, column within_1e-12) and printed."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    x, y, z = grids(nx, ny, nz, stretch)
    sc = (0.7, 1.0, 2.5)[:nscal]
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 23, noise=1e-3)
    ss = [s0[0] * (1.0 + 0.3 * i) + 0.1 * i for i in range(nscal)]
    visc, dt = 1.0 / 5000.0, 1e-3

    make, source = oracle_factory(x, y, z, nscal, visc, sc, stretch)
    o = make()
    for i in range(3):
        o.q[i] = q0[i].copy()
    for i in range(nscal):
        o.s[i] = ss[i].copy()
    kdt, kco = [1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0], [-5.0 / 9.0, -153.0 / 128.0]
    for k in range(3):                      # step 1 on the oracle alone
        o.time_substep(dt * kdt[k], 1.0 if k == 2 else kco[k], k != 2)
    q1, s1 = [a.copy() for a in o.q], [a.copy() for a in o.s]
    sched = [(dt * kdt[k], 1.0 if k == 2 else kco[k], k != 2, k == 0) for k in range(3)]
    B, S = oracle_substeps(("projected", nx, ny, nz), make, q1, s1, sched, nsamples=1)
    d = Dns(x, y, z, nscal=nscal, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=REF_HYPER)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q1[i]))
    for i in range(nscal):
        d.s[i].copy_(torch.from_numpy(s1[i]))
    d.begin_step()
    worst = 0.0
    for k, (dte, kc, scale, _) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kc, scale)
        errs = [rel_err(d.q[i].cpu().numpy(), B[k]["q"][i]) for i in range(3)]
        worst = max(worst, max(errs))
        print("substep %d from a projected field %s: err %s, oracle one-ulp scatter %s" % (k + 4, (nx, ny, nz), ["%.1e" % e for e in errs], ["%.1e" % e for e in S[k]["q"]]))
        check_state(d, B, S, k, names=("q", "s"), tag="projected", source=source)
    print("north-star 1e-12 from a projected field %s: %s (worst %.1e)" % ((nx, ny, nz), "holds" if worst <= FLOOR else "NOT met", worst))


@pytest.mark.parametrize("fuse,nx", [(True, 256), (False, 256), (True, 48)])
def test_full_rk_step_with_begin_step(T, fuse, nx):
    """TIME_RUNGEKUTTA (time.f90:185-298): hq = hs = 0 at the start of the step is a flag here (tlab_dns_begin_step): the arrays may hold
    anything (poisoned with NaN below) and the first operator launch overwrites them.  nx = 48 takes the unfused kernels (memset path)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    ny, nz = 64, 32
    x, y, z = grids(nx, ny, nz, True)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 23)
    d = Dns(x, y, z, nscal=1, visc=1.0 / 800.0, schmidt=(0.7,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=1, visc=1.0 / 800.0, schmidt=(0.7,), yuniform=False)
    d.set_fusion(fuse)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i])); o.q[i] = q0[i].copy()
    d.s[0].copy_(torch.from_numpy(s0[0])); o.s[0] = s0[0].copy()
    for t in d.hq + d.hs:
        t.fill_(float("nan"))
    dtime = 2e-3
    d.TIME_RUNGEKUTTA(dtime)
    for a in o.hq + o.hs:
        a[:] = 0.0
    for k in range(3):
        last = k == 2
        o.time_substep(dtime * d.kdt[k], 1.0 if last else d.kco[k], not last)
    for i in range(3):
        assert rel_err(d.q[i].cpu().numpy(), o.q[i]) <= 1e-12
    assert rel_err(d.s[0].cpu().numpy(), o.s[0]) <= 1e-12


@pytest.mark.parametrize("walls", ["noslip", "freeslip"])
def test_forcing_terms_inside_the_burgers_launches(T, walls):
    """256 x 64 x 64, two scalars, stretched y: every velocity component gets the Burgers term of its own direction last and the one-field launches of
    v along y and w along z add their term of the pressure forcing themselves (k_htile<BURGERS+div>; launch order z, y, x, y, z).  A full Runge-Kutta
    step from NaN-poisoned tendencies: the first launch of each field overwrites (per-field flags: the y launch of u, w, s starts w and continues the
    others), the later substeps accumulate.  Against the oracle at 1e-12, and the launch itself is looked up in the library's own kernel table.
    walls = freeslip (and Neumann scalars), the reference's default walls: the forcing terms still ride on the Burgers launches, v is finished by the
    solver, and every field with a Neumann wall takes one launch along y that forms the wall tendencies of BOUNDARY_BCS_NEUMANN_Y from the finished
    tendency and does the final update (after the gradient was subtracted by the x / z derivative launch itself)."""
    import ctypes
    import torch
    from tlab_amd.dns import Dns, velocity_bcs, scalar_bcs
    from tlab_amd.lib import load
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz, nscal = 256, 64, 64, 2
    x, y, z = grids(nx, ny, nz, True)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 29)
    sc = (0.7, 2.0)
    ss = [s0[0], s0[0] * 0.5 + 0.2]
    d = Dns(x, y, z, nscal=nscal, visc=1.0 / 800.0, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=nscal, visc=1.0 / 800.0, schmidt=sc, yuniform=False)
    if walls == "freeslip":
        d.set_bcs("freeslip", "freeslip", "neumann", "neumann")
        o.flow_jmin, o.flow_jmax = velocity_bcs("freeslip"), velocity_bcs("freeslip")
        o.scal_jmin, o.scal_jmax = [scalar_bcs("neumann")] * nscal, [scalar_bcs("neumann")] * nscal
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i])); o.q[i] = q0[i].copy()
    for i in range(nscal):
        d.s[i].copy_(torch.from_numpy(ss[i])); o.s[i] = ss[i].copy()
    for t in d.hq + d.hs:
        t.fill_(float("nan"))
    L = load()
    L.tlab_profile_reset(); L.tlab_profile_enable(1)
    dtime = 2e-3
    d.TIME_RUNGEKUTTA(dtime)
    torch.cuda.synchronize()
    L.tlab_profile_enable(0)
    buf = ctypes.create_string_buffer(16384); L.tlab_profile_report(buf, 16384)
    rows = {r.split("\t")[0]: int(r.split("\t")[1]) for r in buf.value.decode().splitlines() if "\t" in r}
    assert rows.get("k_htile<BURGERS+div>") == 6, rows          # v along y and w along z, three substeps
    if walls == "noslip":
        assert "k_rtile<P1>" not in rows or rows["k_rtile<P1>"] == 3, rows      # only the gradient-final launches of w are left of that kernel
    else:       # the two scalars: BOUNDARY_BCS_NEUMANN_Y + final update in one launch each, three substeps; u and w: their wall planes from weighted
        # sums (one launch each per substep; the scalars' interior rides on the x Burgers launch) and the gradient kernels finish u and w; no
        # derivative pass along y and no separate update pass is left
        assert "k_rtile<P1+neumann final>" not in rows and rows.get("k_wall_weighted") == 12, rows
        assert "k_final_update" not in rows and "k_sub3" not in rows, rows
    for a in o.hq + o.hs:
        a[:] = 0.0
    for k in range(3):
        last = k == 2
        o.time_substep(dtime * d.kdt[k], 1.0 if last else d.kco[k], not last)
    for i in range(3):
        assert rel_err(d.q[i].cpu().numpy(), o.q[i]) <= 1e-12, i
    for i in range(nscal):
        assert rel_err(d.s[i].cpu().numpy(), o.s[i]) <= 1e-12, i


def test_time_courant_and_dilatation_vs_oracle(T):
    """TIME_COURANT (time.f90:365) and the dilatation monitor (FI_INVARIANT_P + MINMAX, dns_local.f90:157-187)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 64, 48, 32
    x, y, z = grids(nx, ny, nz, True)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 13)
    d = Dns(x, y, z, nscal=1, visc=1.0 / 700.0, schmidt=(0.5,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    o = DnsOracle(x, y, z, nscal=1, visc=1.0 / 700.0, schmidt=(0.5,), yuniform=False)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i])); o.q[i] = q0[i].copy()
    (p1, p2), dt = d.TIME_COURANT(1.2, 0.25)
    (r1, r2), rdt = o.time_courant(1.2, 0.25)
    assert abs(p1 - r1) <= 1e-14 * abs(r1) and abs(p2 - r2) <= 1e-14 * abs(r2) and abs(dt - rdt) <= 1e-14 * rdt
    dmin, dmax = d.dilatation_bounds()
    ref = -o.fi_invariant_p()
    assert abs(dmin - ref.min()) <= 1e-12 * np.abs(ref).max() and abs(dmax - ref.max()) <= 1e-12 * np.abs(ref).max()
    assert rel_err(d.txc[0][: d.n].cpu().numpy(), -ref) <= 1e-13


def test_projection_makes_interior_divergence_vanish(T):
    """max|div(u/dte + hq)| * dte ~ 2e-15 in the interior after the pressure correction."""
    import torch
    from tlab_amd.dns import Dns
    nx, ny, nz = 128, 96, 64
    x, y, z = grids(nx, ny, nz, True)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 5)
    d = Dns(x, y, z, nscal=1, visc=1e-3, schmidt=(1.0,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    dte = 1e-3
    d.RHS_GLOBAL_INCOMPRESSIBLE_1(dte)
    # the wall planes of hq are overwritten by the BCs after the projection (:373-375); the compact y-derivative spreads that
    # jump into the domain with a decay of ~0.38 per row, so the invariant is checked > 30 rows away from the walls
    a, b, c, div = (torch.empty_like(d.q[0]) for _ in range(4))
    div.zero_()
    for dirn, part in ((1, T.OPR_Partial_X), (2, T.OPR_Partial_Y), (3, T.OPR_Partial_Z)):
        a.copy_(d.q[dirn - 1] / dte + d.hq[dirn - 1])
        part(T.OPR_P1, nx, ny, nz, 0, d.g[dirn - 1], a, b)
        div += b
    scale = float((d.q[0].abs().max() / dte))
    interior = div.view(nz, ny, nx)[:, 36:-36, :]
    assert float(interior.abs().max()) / scale * (x[1] - x[0]) <= 1e-11


def test_full_size_substep_properties(T):
    """BASELINE configs[2] size (512^3, one scalar), where the oracle is out of reach: (1) the fused driver (multi-field Burgers launches,
    operand / gradient / update fusions, chunked ODE kernel, own z-FFT) and the literal operator sequence of the reference (set_fusion(False))
    must agree to round-off after a full RK3 step; (2) the projected velocity is solenoidal in the interior."""
    import torch
    from tlab_amd.dns import Dns
    n = 512
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(512)
    X = torch.arange(n, dtype=torch.float64, device="cuda").view(1, 1, n) / n
    Y = torch.arange(n, dtype=torch.float64, device="cuda").view(1, n, 1) / (n - 1)
    Z = torch.arange(n, dtype=torch.float64, device="cuda").view(n, 1, 1) / n
    wall = torch.sin(np.pi * Y)
    two_pi = 2 * np.pi
    shapes = [torch.sin(two_pi * X) * torch.cos(2 * two_pi * Y) * torch.sin(3 * two_pi * Z), torch.cos(two_pi * X) * torch.sin(two_pi * Y) * torch.sin(2 * two_pi * Z),
              torch.sin(2 * two_pi * X) * torch.cos(two_pi * Y) * torch.cos(two_pi * Z), torch.cos(3 * two_pi * X) * torch.cos(two_pi * Y) * torch.sin(two_pi * Z)]
    fields = [((sh + 0.1 * (2 * torch.rand(n, n, n, dtype=torch.float64, device="cuda", generator=gen) - 1)) * wall).reshape(-1) for sh in shapes]
    out = []
    for fuse in (True, False):
        d = Dns(x, y, x.copy(), nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=REF_HYPER)
        d.set_fusion(fuse)
        for t, f in zip(d.q + d.s, fields):
            t.copy_(f)
        d.TIME_RUNGEKUTTA(1e-3)
        out.append([t.clone() for t in d.q + d.s])
        if fuse:
            # interior divergence of the new velocity (the last substep's projection), cf. test_projection_makes_interior_divergence_vanish
            dmin, dmax = None, None
            d.FI_INVARIANT_P(d.txc[0], d.txc[1])
            div = d.txc[0][: d.n].view(n, n, n)[:, 40:-40, :]
            scale = float(d.q[0].abs().max()) * n
            assert float(div.abs().max()) / scale <= 1e-10
        del d
        torch.cuda.empty_cache()
    for a, b in zip(*out):
        assert float((a - b).abs().max() / b.abs().max()) <= 1e-12


@pytest.mark.parametrize("stretch", [False, True])
def test_hundred_steps_of_decaying_flow_stay_bounded(T, stretch):
    """What the domain offers at any size: without forcing the kinetic energy of the flow between no-slip walls decays, the scalar stays inside
    its initial bounds (up to the dispersion of the scheme) and the dilatation stays at the level the projection leaves.  100 Runge-Kutta
    steps (300 substeps, each with its Poisson solve) with the time step of TIME_COURANT, consistent wall closure of the second derivative
    (hyper_bc1_ext = 0: the value the reference reads past its coefficient array is unstable over this many steps on a stretched grid,
    DESIGN.md section 2)."""
    import torch
    from tlab_amd.dns import Dns
    nx, ny, nz = 128, 96, 64
    x, y, z = grids(nx, ny, nz, stretch)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 31)
    d = Dns(x, y, z, nscal=1, visc=1.0 / 1500.0, schmidt=(1.0,), yuniform=not stretch, hyper_bc1_ext=0.0)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))

    def energy():
        return float(sum((t * t).sum() for t in d.q))

    e_hist, dil = [energy()], []
    smin0, smax0 = float(d.s[0].min()), float(d.s[0].max())
    for step in range(100):
        _, dt = d.TIME_COURANT(1.2, 0.25)
        assert 0.0 < dt < 1.0
        d.TIME_RUNGEKUTTA(dt)
        if step % 10 == 9:
            e_hist.append(energy())
            dmin, dmax = d.dilatation_bounds()
            dil.append(max(abs(dmin), abs(dmax)))
    assert all(bool(torch.isfinite(t).all()) for t in d.q + d.s)
    assert all(b < a for a, b in zip(e_hist, e_hist[1:])), e_hist                  # monotone viscous decay
    assert e_hist[-1] > 0.02 * e_hist[0]                                           # ... of a flow that is still there
    span = smax0 - smin0
    assert float(d.s[0].min()) >= smin0 - 0.05 * span and float(d.s[0].max()) <= smax0 + 0.05 * span
    (p1, _), _ = d.TIME_COURANT(1.2, 0.25)                                         # p1 = max(|u_i| / h_i): the size of the terms of div(q)
    assert max(dil) <= 0.05 * max(p1, 1e-300) or max(dil) <= dil[0] * 2.0, (dil, p1)


# ---------------------------------------------------------------------------------------------------------------------------------
# nse_eqns = anelastic: the density weights of the RHS and of OPR_Burgers (rhs_global_incompressible_1.f90:211-214, 275-277, 326-329;
# opr_burgers.f90:128-183, 504-507) with given background profiles
# ---------------------------------------------------------------------------------------------------------------------------------
def background(y):
    rb = 1.0 + 0.6 * np.exp(-2.5 * (y - y[0]) / (y[-1] - y[0]))          # a density decreasing with height
    return rb, 1.0 / rb


@pytest.mark.parametrize("nx,ny,nz,stretch", [(32, 40, 16, True), (64, 64, 64, False), (256, 64, 32, True)])
def test_anelastic_burgers_operators_vs_oracle(T, nx, ny, nz, stretch):
    """OPR_Burgers_X/Y/Z with rhoinv active: generic kernels at (32, 40, 16), the fast derivative kernels at the other sizes."""
    import torch
    from oracle import tlab_oracle as O
    from tlab_amd.lib import load, check
    x, y, z = grids(nx, ny, nz, stretch)
    rb, ri = background(y)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 11)
    dp = __import__("ctypes").POINTER(__import__("ctypes").c_double)
    check(load().tlab_opr_burgers_set_anelastic(ny, rb.ctypes.data_as(dp), ri.ctypes.data_as(dp)), "set_anelastic")
    try:
        s_, v_ = torch.from_numpy(s0[0]).cuda(), torch.from_numpy(q0[1]).cuda()
        res, tmp = torch.empty_like(s_), torch.empty_like(s_)
        for d, (nodes, per, uni) in {1: (x, True, True), 2: (y, False, not stretch), 3: (z, True, True)}.items():
            gp, go = T.FdmPlan(nodes, per, uni, hyper_bc1_ext=REF_HYPER), O.FdmPlan(nodes, per, uni)
            burg = (T.OPR_Burgers_X, T.OPR_Burgers_Y, T.OPR_Burgers_Z)[d - 1]
            burg(T.OPR_B_U_IN, 1.0 / 300.0, nx, ny, nz, 0, gp, s_, v_, res, tmp)
            ref = O.opr_burgers(d, nx, ny, nz, 0, go, 1.0 / 300.0, s0[0], q0[1], anelastic=(rb, ri))[0]
            assert rel_err(res.cpu().numpy(), ref) <= 1e-12, d
            plain = O.opr_burgers(d, nx, ny, nz, 0, go, 1.0 / 300.0, s0[0], q0[1])[0]
            assert rel_err(ref, plain) > 1e-3          # (the weights do something)
    finally:
        check(load().tlab_opr_burgers_set_anelastic(0, None, None), "set_anelastic off")
    with pytest.raises(T.TlabError):                   # ribackground must be the reciprocal profile
        check(load().tlab_opr_burgers_set_anelastic(ny, rb.ctypes.data_as(dp), rb.ctypes.data_as(dp)), "set_anelastic")


@pytest.mark.gpu_extra
@pytest.mark.parametrize("bcs", ["noslip", "freeslip", "noslip, set through the operators"])
@pytest.mark.parametrize("nx,ny,nz,stretch", [(32, 40, 16, True), (128, 64, 64, True), (256, 64, 64, True)])      # the last: the fused Burgers launches
def test_anelastic_substep_vs_oracle(T, nx, ny, nz, stretch, bcs):
    """"set through the operators": only tlab_opr_burgers_set_anelastic is called (what the Fortran shim OPR_Burgers_AMD_Anelastic does) -- the RHS
    driver follows the operator state, so the Burgers terms and the density weights of the pressure step cannot disagree about the equations."""
    import ctypes
    import torch
    from tlab_amd.dns import Dns, velocity_bcs
    from tlab_amd.lib import load, check
    from oracle.tlab_oracle_rhs import DnsOracle
    x, y, z = grids(nx, ny, nz, stretch)
    rb, ri = background(y)
    visc, sc = 1.0 / 800.0, (0.7,)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 5)
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=REF_HYPER)
    if bcs.endswith("operators"):
        dp = ctypes.POINTER(ctypes.c_double)
        rbc, ric = np.ascontiguousarray(rb), np.ascontiguousarray(ri)
        check(load().tlab_opr_burgers_set_anelastic(ny, rbc.ctypes.data_as(dp), ric.ctypes.data_as(dp)), "tlab_opr_burgers_set_anelastic")
        bcs = "noslip"
    else:
        d.set_anelastic(rb)
    try:
        if bcs == "freeslip":
            d.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
        for i in range(3):
            d.q[i].copy_(torch.from_numpy(q0[i]))
        d.s[0].copy_(torch.from_numpy(s0[0]))
        dtime = 2e-3
        sched = [(dtime * d.kdt[k], d.kco[k], True) for k in range(2)]

        def make_oracle():
            o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, anelastic=(rb, ri))
            if bcs == "freeslip":
                o.flow_jmin = o.flow_jmax = velocity_bcs("freeslip"); o.scal_jmin, o.scal_jmax = [4], [3]
            return o
        B, S = oracle_substeps(("anel", nx, ny, nz, stretch, bcs), make_oracle, q0, s0, sched, nsamples=2)
        other = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch, hyper_bc1_ext=REF_HYPER)
        del other                      # destroying a driver that did not switch the state on must leave it on
        for k, (dte, kco, scale) in enumerate(sched):
            d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
            check_state(d, B, S, k, tag="anelastic " + bcs)
        # and it is not the incompressible result
        o0 = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=not stretch)
        for i in range(3):
            o0.q[i] = q0[i].copy()
        o0.s[0] = s0[0].copy()
        o0.time_substep(*sched[0])
        assert rel_err(B[0]["q"][1], o0.q[1]) > 1e-6
    finally:
        d.set_anelastic(None)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("sides", ["bottom", "both"])
def test_dynamic_surface_bcs_vs_oracle(T, sides, fuse):
    """Scalar1SfcTypeJmin = linear (examples/Case88): BOUNDARY_BCS_SURFACE_Y on the device against the oracle, three substeps (the kept tendency
    plane of one substep feeds the next)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 64, 48, 32
    x, y, z = grids(nx, ny, nz, True)
    visc, sc = 1.0 / 800.0, (0.7,)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 13)
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    d.set_fusion(fuse)
    jmax = "linear" if sides == "both" else "static"
    d.set_surface_bcs(["linear"], [jmax], [0.35], [-0.2])
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    dtime = 2e-3
    sched = [(dtime * d.kdt[k], d.kco[k] if k < 2 else 1.0, k < 2) for k in range(3)]      # one full RK3 step (time.f90:220-298)

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.sfc_jmin, o.cpl_jmin = [1], [0.35]
        o.sfc_jmax, o.cpl_jmax = [1 if sides == "both" else 0], [-0.2]
        return o
    B, S = oracle_substeps(("sfc", sides), make_oracle, q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, tag="surface " + sides)
    assert np.abs(B[2]["hs"][0].reshape(nz, ny, nx)[:, 0, :]).max() > 1e-3          # (the bottom plane of the tendency is alive)


@pytest.mark.parametrize("scheme1,nx,ny,nz", [(5, 64, 40, 32), (5, 256, 64, 64), (4, 64, 36, 32)])
def test_substep_with_other_first_derivatives_and_the_default_elliptic_solver(T, scheme1, nx, ny, nz):
    """SpaceOrder1 = CompactJacobian6Penta (5) / CompactJacobian4 (4) with the DEFAULT EllipticOrder: the factorized Poisson solver then inverts a
    first derivative with (5, 7) / (3, 3) diagonals, i.e. 7- / 3-diagonal integral systems (FDM_Int1 with HEPTADFS / TRIDFS, fdm_integral.f90:75-83)
    -- the last piece of SURVEY 8f n3.  Two substeps against the oracle within the scatter bound."""
    import torch
    from tlab_amd.dns import Dns
    from oracle import tlab_oracle as O
    from oracle.tlab_oracle_rhs import DnsOracle
    x, y, z = grids(nx, ny, nz, True)
    visc, sc = 1.0 / 800.0, (0.7,)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 31)
    gp = [T.FdmPlan(x, True, True, scheme1, 7, hyper_bc1_ext=REF_HYPER), T.FdmPlan(y, False, False, scheme1, 7, hyper_bc1_ext=REF_HYPER),
          T.FdmPlan(z, True, True, scheme1, 7, hyper_bc1_ext=REF_HYPER)]
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, plans=gp, hyper_bc1_ext=REF_HYPER)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    sched = [(2e-3 * d.kdt[k], d.kco[k], True) for k in range(2)]

    def make_oracle():
        go = [O.FdmPlan(x, True, True, scheme1, 7), O.FdmPlan(y, False, False, scheme1, 7), O.FdmPlan(z, True, True, scheme1, 7)]
        return DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, plans=go)
    B, S = oracle_substeps(("scheme1", scheme1, nx, ny, nz), make_oracle, q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, tag="scheme1 = %d" % scheme1)


@pytest.mark.parametrize("fuse", [True, False])
def test_dynamic_surface_bcs_over_two_runge_kutta_steps(T, fuse):
    """The kept tendency planes of the surface model across a step boundary: the device driver is TOLD that hq, hs are zero at the start of a step
    (begin_step instead of the fill of time.f90:212-216), so the planes it keeps in the first substep of the SECOND step must be zero too, not
    what hs still holds from the step before (BcsScal%ref = 0 there in the reference)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 64, 48, 32
    x, y, z = grids(nx, ny, nz, True)
    visc, sc = 1.0 / 800.0, (0.7,)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 13)
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    d.set_fusion(fuse)
    d.set_surface_bcs(["linear"], ["linear"], [0.35], [-0.2])
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    dtime = 2e-3
    sched = [(dtime * d.kdt[k % 3], d.kco[k % 3] if k % 3 < 2 else 1.0, k % 3 < 2, k % 3 == 0) for k in range(6)]

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.sfc_jmin, o.cpl_jmin, o.sfc_jmax, o.cpl_jmax = [1], [0.35], [1], [-0.2]
        return o
    B, S = oracle_substeps(("sfc two steps",), make_oracle, q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale, new_step) in enumerate(sched):
        if new_step:
            d.begin_step()
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, tag="surface, two steps")
    # the planes the second step would wrongly keep are not small: the unscaled last-substep tendency of step 1
    assert np.abs(B[2]["hs"][0].reshape(nz, ny, nx)[:, 0, :]).max() > 1e-3


@pytest.mark.parametrize("fuse", [True, False])
def test_without_remove_divergence_vs_oracle(T, fuse):
    """dns.ini remove_divergence = none: the forcing of the pressure is div(hq) alone (rhs_global_incompressible_1.f90:234-250)."""
    import torch
    from tlab_amd.dns import Dns
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 64, 40, 32
    x, y, z = grids(nx, ny, nz, True)
    visc, sc = 1.0 / 800.0, (0.7,)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 17)
    d = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    d.set_fusion(fuse)
    d.set_remove_divergence(False)
    for i in range(3):
        d.q[i].copy_(torch.from_numpy(q0[i]))
    d.s[0].copy_(torch.from_numpy(s0[0]))
    sched = [(2e-3 * d.kdt[k], d.kco[k], True) for k in range(2)]

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.remove_divergence = False
        return o
    B, S = oracle_substeps(("nodiv",), make_oracle, q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        check_state(d, B, S, k, tag="remove_divergence off")
