"""The NATIVE z-slab driver (tlab_amd/csrc/slab.cpp behind tlab_slab_dns_*, what a Fortran / MPI host and `bench.py --gpus N` run) on ONE device:
all npro_k ranks inside this process through the loopback transport.  It must reproduce the Python driver it was ported from
(tlab_amd/parallel.py::SlabDns, itself held to the single domain and to the oracle in tests/test_gpu_slab.py) TO THE BIT -- same kernels, same
arguments, same order per rank -- and, independently, the single-domain substep and the oracle within the scatter bound."""
import numpy as np
import pytest
from scatter import substep_scatter, bound, ref_of
import cases as C

REF_HYPER = 0.1      # wall closure of the flang-built reference (defect 1)

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


def _fields(x, y, z, seed):
    rng = np.random.default_rng(seed)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    return [((np.sin(np.pi * X + k) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(5)]


def _grid(nx, ny, nz):
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    return x, y, z


@pytest.mark.parametrize("P,nx,ny,nz,bcs,ns,stages", [
    (2, 32, 16, 128, "noslip", 1, "2"), (2, 32, 16, 128, "noslip", 1, "1"), (4, 64, 24, 256, "freeslip", 1, "2"), (8, 32, 16, 512, "noslip", 1, "2"),
    (3, 48, 16, 192, "noslip", 2, "2"), (2, 32, 16, 128, "neumann-scalars", 2, "2"), (2, 32, 16, 128, "noslip", 0, "2")])
def test_native_driver_is_bit_identical_to_the_python_driver(T, P, nx, ny, nz, bcs, ns, stages, monkeypatch):
    import torch
    from tlab_amd.parallel import SlabDns, LoopbackComm
    from tlab_amd.slab import NativeSlabDns
    monkeypatch.setenv("TLAB_PENCIL_STAGES", stages)
    x, y, z = _grid(nx, ny, nz)
    f = _fields(x, y, z, P + ns)
    kw = dict(nscal=ns, visc=1.0 / 600.0, schmidt=(0.8, 1.3)[:ns], yuniform=False, hyper_bc1_ext=REF_HYPER)
    py = SlabDns(LoopbackComm(P), x, y, z, zmode="halo", **kw)
    nat = NativeSlabDns("loopback", x, y, z, size=P, fused_x=False, **kw)     # separate repack passes + rocFFT's inverse, like the Python driver
    assert py.stages == nat.stages == int(stages) and nat.kmax == py.kmax and not nat.fused_x
    walls = {"noslip": (), "freeslip": ("freeslip", "freeslip", "neumann", "dirichlet"), "neumann-scalars": ("noslip", "freeslip", "neumann", "neumann")}[bcs]
    if walls:
        py.set_bcs(*walls); nat.set_bcs(*walls)
    for d in (py, nat):
        for i in range(3):
            d.scatter("q", i, torch.from_numpy(f[i]).cuda())
        for i in range(ns):
            d.scatter("s", i, torch.from_numpy(f[3 + i]).cuda())
    for k in range(4):          # one full Runge-Kutta step and the first substep of the next (fresh tendencies again)
        py.substep_of_cycle(k, 2e-3)
        nat.substep_of_cycle(k, 2e-3)
    torch.cuda.synchronize()
    for name in ("q", "s", "hq", "hs"):
        for r in range(P):
            for i, (a, b) in enumerate(zip(py.st[r][name], nat.st[r][name])):
                assert bool(torch.isfinite(b).all()) and float(b.abs().max()) > 0.0
                assert torch.equal(a, b), (name, r, i, float((a - b).abs().max()))
    # the RHS-only entry (what an unpatched time.f90 calls before its own update loops)
    py.RHS_GLOBAL_INCOMPRESSIBLE_1(1e-3); nat.RHS_GLOBAL_INCOMPRESSIBLE_1(1e-3)
    for name in ("hq", "hs"):
        for r in range(P):
            for a, b in zip(py.st[r][name], nat.st[r][name]):
                assert torch.equal(a, b), (name, r)
    # monitors: TIME_COURANT with its MPI_MAX (time.f90:522), dilatation bounds of DNS_BOUNDS_CONTROL
    assert py.TIME_COURANT(1.2, 0.3) == nat.TIME_COURANT(1.2, 0.3)
    assert py.dilatation_bounds() == nat.dilatation_bounds()


def test_redrawn_arrays_give_the_same_run_bit_for_bit(T):
    """NativeSlabDns.redraw_arrays (what bench.py does before its timed region on slabs): the ranks' arrays move to other allocations, the fields come
    along, and the run is the run without it."""
    import torch
    from tlab_amd.slab import NativeSlabDns
    P, nx, ny, nz = 4, 64, 24, 256
    x, y, z = _grid(nx, ny, nz)
    f = _fields(x, y, z, 11)
    kw = dict(nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    a, b = NativeSlabDns("loopback", x, y, z, size=P, **kw), NativeSlabDns("loopback", x, y, z, size=P, **kw)
    for d in (a, b):
        for i in range(3):
            d.scatter("q", i, torch.from_numpy(f[i]).cuda())
        d.scatter("s", 0, torch.from_numpy(f[3]).cuda())
    a.substep_of_cycle(0, 2e-3); b.substep_of_cycle(0, 2e-3)          # mid-step: the tendencies carry values too
    rep = b.redraw_arrays(pool=30, seed=5)
    assert rep["pool"] == 30
    ptrs = [t.data_ptr() for r in range(P) for name in ("q", "s", "hq", "hs", "txc") for t in b.st[r][name]]
    assert len(set(ptrs)) == P * 17
    for k in range(1, 5):
        a.substep_of_cycle(k, 2e-3); b.substep_of_cycle(k, 2e-3)
    torch.cuda.synchronize()
    for name in ("q", "s", "hq", "hs"):
        for r in range(P):
            for u, v in zip(a.st[r][name], b.st[r][name]):
                assert bool(torch.isfinite(v).all()) and torch.equal(u, v), (name, r)


@pytest.mark.parametrize("bcs,fused,stages", [("freeslip", True, "2"), ("noslip", True, "2"), ("noslip", True, "1"), ("noslip", False, "2")])
def test_native_driver_equals_single_domain_and_oracle(T, bcs, fused, stages, monkeypatch):
    """fused: the repack passes folded into the own x-transforms, v finished by the inverse transform of dp^/dy (no-slip walls)."""
    import torch
    monkeypatch.setenv("TLAB_PENCIL_STAGES", stages)
    from tlab_amd.dns import Dns, velocity_bcs
    from tlab_amd.slab import NativeSlabDns
    from oracle.tlab_oracle_rhs import DnsOracle
    P, nx, ny, nz = 4, 128, 24, 256          # nx >= 128: the library's own x-transforms apply
    case = C.slab_native(bcs)
    x, y, z, visc, sc = (case[k] for k in ("x", "y", "z", "visc", "sc"))
    f = case["q0"] + case["s0"]
    one = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    nat = NativeSlabDns("loopback", x, y, z, size=P, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER, fused_x=fused)
    assert nat.fused_x == fused and nat.stages == int(stages)
    walls = ("freeslip", "freeslip", "neumann", "dirichlet") if bcs == "freeslip" else ("noslip", "noslip", "dirichlet", "dirichlet")
    one.set_bcs(*walls)
    nat.set_bcs(*walls)
    for i in range(3):
        t = torch.from_numpy(f[i]).cuda()
        one.q[i].copy_(t); nat.scatter("q", i, t)
    t = torch.from_numpy(f[3]).cuda()
    one.s[0].copy_(t); nat.scatter("s", 0, t)
    dtime = 2e-3
    for k in range(2):
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)
        nat.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.flow_jmin = o.flow_jmax = velocity_bcs(bcs)
        if bcs == "freeslip":
            o.scal_jmin, o.scal_jmax = [4], [3]
        return o
    assert [(dtime * one.kdt[k], one.kco[k], True) for k in range(2)] == case["sched"]
    if ("native", bcs) not in _ORACLE:          # (the fused / stages parametrizations share one oracle run)
        _ORACLE[("native", bcs)] = substep_scatter(make_oracle, f[:3], f[3:4], case["sched"], nsamples=3)
    B, S = _ORACLE[("native", bcs)]
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s), ("hs", one.hs)):
        for i, rf in enumerate(ref):
            got = torch.cat([nat.st[r][name][i] for r in range(P)])
            tol = bound(S[1][name][i], ref=ref_of(case["key"], 1, name, i))
            err = float((got - rf).abs().max() / rf.abs().max())
            assert err <= tol, ("native slabs vs single domain", name, i, err, tol)
            ob = torch.from_numpy(B[1][name][i]).cuda()
            err = float((got - ob).abs().max() / ob.abs().max())
            assert err <= tol, ("native slabs vs oracle", name, i, err, tol)


def test_native_driver_with_the_direct_schemes_is_bit_identical(T):
    """The scheme set of examples/Case81-93 (SpaceOrder2 = CompactDirect6 in y, EllipticOrder = CompactDirect6) through the native driver."""
    import os
    import torch
    from tlab_amd.parallel import SlabDns, LoopbackComm
    from tlab_amd.slab import NativeSlabDns
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct_y.npz"))
    P, nx, ny, nz = 2, 32, 64, 128
    tab = {k[len("ny%d_" % ny):]: G[k] for k in G.files if k.startswith("ny%d_" % ny)}
    x, y, z = np.arange(nx) / nx * 2.0, tab["nodes"], np.arange(nz) / nz
    mk = lambda: [T.FdmPlan(x, True, True), T.FdmPlan.from_tables(tab, False, T.FDM_COM6_JACOBIAN, T.FDM_COM6_DIRECT), T.FdmPlan(z, True, True)]   # noqa: E731
    g1, g2 = mk(), mk()
    kw = dict(nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    py = SlabDns(LoopbackComm(P), x, y, z, zmode="halo", plans=g1, gy_elliptic=g1[1], **kw)
    nat = NativeSlabDns("loopback", x, y, z, size=P, plans=g2, gy_elliptic=g2[1], fused_x=False, **kw)
    f = _fields(x, y, z, 5)
    for d in (py, nat):
        for i in range(4):
            d.scatter("q" if i < 3 else "s", i if i < 3 else 0, torch.from_numpy(f[i]).cuda())
    for k in range(3):
        py.substep_of_cycle(k, 2e-3); nat.substep_of_cycle(k, 2e-3)
    for name in ("q", "s", "hq", "hs"):
        for r in range(P):
            for a, b in zip(py.st[r][name], nat.st[r][name]):
                assert float(b.abs().max()) > 0.0 and torch.equal(a, b), (name, r)


@pytest.mark.parametrize("walls", [("freeslip", "freeslip", "neumann", "dirichlet"), ("freeslip", "noslip", "neumann", "neumann")])
def test_native_driver_default_walls_on_the_wall_plane_route(T, walls, monkeypatch):
    """The reference's default walls (free-slip u, w; Neumann scalars) in the slab driver WITHOUT a derivative pass per Neumann field: the fields are
    finished with zero wall tendencies by the kernels that hold their last term and k_wall_weighted + k_wall_fix set the wall planes (local in y, no
    exchange).  Against the derivative-pass route (TLAB_NEUMANN_PLANES=0) and the single domain, within the composed-path bound; the route must
    actually have been taken (k_wall_weighted launches in the library's profile)."""
    import ctypes
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.lib import load
    from tlab_amd.slab import NativeSlabDns
    from oracle.tlab_oracle_rhs import DnsOracle
    from tlab_amd.dns import velocity_bcs
    L = load()
    P, nx, ny, nz = 2, 128, 64, 128
    x, y, z = _grid(nx, ny, nz)
    f = _fields(x, y, z, 29)
    visc, sc = 1.0 / 600.0, (0.8,)
    kw = dict(nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    one = Dns(x, y, z, **kw)
    one.set_bcs(*walls)
    runs = {}
    for tag, env in (("planes", None), ("derivative", "0")):
        if env is None:
            monkeypatch.delenv("TLAB_NEUMANN_PLANES", raising=False)
        else:
            monkeypatch.setenv("TLAB_NEUMANN_PLANES", env)
        nat = NativeSlabDns("loopback", x, y, z, size=P, fused_x=True, **kw)
        nat.set_bcs(*walls)
        for i in range(4):
            nat.scatter("q" if i < 3 else "s", i if i < 3 else 0, torch.from_numpy(f[i]).cuda())
        L.tlab_profile_reset(); L.tlab_profile_enable(1)
        for k in range(2):
            nat.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(2e-3 * one.kdt[k], one.kco[k], True)
        torch.cuda.synchronize()
        L.tlab_profile_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        L.tlab_profile_report(buf, len(buf))
        took = b"k_wall_weighted" in buf.value
        assert took == (tag == "planes"), (tag, buf.value.decode()[:400])
        runs[tag] = {name: [torch.cat([nat.st[r][name][i] for r in range(P)]) for i in range(len(nat.st[0][name]))] for name in ("q", "hq", "s", "hs")}
        nat.close()
    monkeypatch.delenv("TLAB_NEUMANN_PLANES", raising=False)
    for i in range(4):
        (one.q[i] if i < 3 else one.s[0]).copy_(torch.from_numpy(f[i]).cuda())
    for k in range(2):
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(2e-3 * one.kdt[k], one.kco[k], True)

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.flow_jmin, o.flow_jmax = velocity_bcs(walls[0]), velocity_bcs(walls[1])
        o.scal_jmin, o.scal_jmax = [4 if walls[2] == "neumann" else 3], [4 if walls[3] == "neumann" else 3]
        return o
    B, S = substep_scatter(make_oracle, f[:3], f[3:4], [(2e-3 * one.kdt[k], one.kco[k], True) for k in range(2)], nsamples=2)
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s), ("hs", one.hs)):
        for i, rf in enumerate(ref):
            tol = bound(S[1][name][i])
            ob = torch.from_numpy(B[1][name][i]).cuda()
            for tag in ("planes", "derivative"):
                got = runs[tag][name][i]
                assert float((got - rf).abs().max() / rf.abs().max()) <= tol, (tag, "vs single domain", name, i)
                assert float((got - ob).abs().max() / ob.abs().max()) <= tol, (tag, "vs oracle", name, i)


@pytest.mark.parametrize("P,sides,walls", [(2, "bottom", "noslip"), (4, "both", "freeslip")])
def test_native_driver_dynamic_surface_model_on_slabs(T, P, sides, walls):
    """Scalar1SfcTypeJmin/Jmax = linear (examples/Case88) on z-slabs: BOUNDARY_BCS_SURFACE_Y needs the plane average of d s / dy over ALL ranks (AVG1V2D,
    boundary_bcs.f90:520) -- the transport's all-reduce with MPI_SUM.  Four substeps across a step boundary (the kept tendency plane of one substep feeds
    the next; zeroed at a step start) against the single domain and the oracle."""
    import torch
    from tlab_amd.dns import Dns, velocity_bcs
    from tlab_amd.slab import NativeSlabDns
    from oracle.tlab_oracle_rhs import DnsOracle
    nx, ny, nz = 128, 48, 64 * P
    x, y, z = _grid(nx, ny, nz)
    f = _fields(x, y, z, 31)
    visc, sc = 1.0 / 600.0, (0.7,)
    kw = dict(nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    one = Dns(x, y, z, **kw)
    nat = NativeSlabDns("loopback", x, y, z, size=P, **kw)
    jmax = "linear" if sides == "both" else "static"
    bc = (walls, walls, "dirichlet", "dirichlet")
    for d in (one, nat):
        d.set_bcs(*bc)
        d.set_surface_bcs(["linear"], [jmax], [0.35], [-0.2])
    for i in range(4):
        t = torch.from_numpy(f[i]).cuda()
        (one.q[i] if i < 3 else one.s[0]).copy_(t)
        nat.scatter("q" if i < 3 else "s", i if i < 3 else 0, t)
    dtime = 2e-3
    sched = [(dtime * one.kdt[k % 3], one.kco[k % 3] if k % 3 < 2 else 1.0, k % 3 < 2, k % 3 == 0) for k in range(4)]
    one.begin_step()
    for k in range(4):
        if k == 3:
            one.begin_step()
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(*sched[k][:3])
        nat.substep_of_cycle(k, dtime)

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.flow_jmin = o.flow_jmax = velocity_bcs(walls)
        o.sfc_jmin, o.cpl_jmin = [1], [0.35]
        o.sfc_jmax, o.cpl_jmax = [1 if sides == "both" else 0], [-0.2]
        return o
    B, S = substep_scatter(make_oracle, f[:3], f[3:4], sched, nsamples=1)
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s), ("hs", one.hs)):
        for i, rf in enumerate(ref):
            got = torch.cat([nat.st[r][name][i] for r in range(P)])
            tol = bound(S[3][name][i])
            assert float((got - rf).abs().max() / rf.abs().max()) <= tol, ("slabs vs single domain", name, i)
            ob = torch.from_numpy(B[3][name][i]).cuda()
            assert float((got - ob).abs().max() / ob.abs().max()) <= tol, ("slabs vs oracle", name, i)
    assert np.abs(B[3]["hs"][0].reshape(nz, ny, nx)[:, 0, :]).max() > 1e-3          # (the bottom plane of the tendency is alive)
    nat.close()


def test_native_driver_refuses_what_it_does_not_build(T):
    """A decomposed run must integrate the same equations as the same tlab.ini on one rank, or stop: with the anelastic operator state
    on, the z-slab driver refuses to be created and refuses to run -- TLAB_EUNSUPPORTED with a message, nothing silently dropped."""
    import ctypes
    import torch
    from tlab_amd.lib import load
    from tlab_amd.slab import NativeSlabDns
    L = load()
    nx, ny, nz = 32, 16, 128
    x, y, z = _grid(nx, ny, nz)
    rb = np.linspace(1.0, 0.8, ny)
    ri = 1.0 / rb
    dp = ctypes.POINTER(ctypes.c_double)
    nat = NativeSlabDns("loopback", x, y, z, size=2, hyper_bc1_ext=REF_HYPER)
    f = _fields(x, y, z, 3)
    for i in range(4):
        nat.scatter("q" if i < 3 else "s", i if i < 3 else 0, torch.from_numpy(f[i]).cuda())
    nat.substep_of_cycle(0, 1e-3)                       # fine while the state is incompressible
    assert L.tlab_opr_burgers_set_anelastic(ny, rb.ctypes.data_as(dp), ri.ctypes.data_as(dp)) == 0
    try:
        with pytest.raises(T.TlabError, match="anelastic"):
            nat.substep_of_cycle(1, 1e-3)
        with pytest.raises(T.TlabError, match="anelastic"):
            NativeSlabDns("loopback", x, y, z, size=2, hyper_bc1_ext=REF_HYPER)
    finally:
        assert L.tlab_opr_burgers_set_anelastic(0, None, None) == 0
    nat.substep_of_cycle(1, 1e-3)                       # and fine again
    nat.close()


def test_native_driver_without_remove_divergence_equals_the_single_domain(T):
    """dns.ini [Main] TermDivergence = no (rhs_global_incompressible_1.f90:234-250: forcing div(hq) alone) is forwarded to the z-slab driver
    (tlab_slab_dns_set_remove_divergence), not ignored: slabs = single domain within the composed-path bound, and the switch changes the result."""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.slab import NativeSlabDns
    from oracle.tlab_oracle_rhs import DnsOracle
    P, nx, ny, nz = 2, 64, 24, 128
    x, y, z = _grid(nx, ny, nz)
    f = _fields(x, y, z, 17)
    visc, sc = 1.0 / 600.0, (0.8,)
    one = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    nat = NativeSlabDns("loopback", x, y, z, size=P, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    on = NativeSlabDns("loopback", x, y, z, size=P, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    one.set_remove_divergence(False)
    nat.set_remove_divergence(False)
    for i in range(4):
        t = torch.from_numpy(f[i]).cuda()
        (one.q[i] if i < 3 else one.s[0]).copy_(t)
        for d in (nat, on):
            d.scatter("q" if i < 3 else "s", i if i < 3 else 0, t)
    dtime = 2e-3
    for k in range(2):
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)
        nat.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)
        on.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.remove_divergence = False
        return o
    B, S = substep_scatter(make_oracle, f[:3], f[3:4], [(dtime * one.kdt[k], one.kco[k], True) for k in range(2)], nsamples=2)
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s)):
        for i, rf in enumerate(ref):
            got = torch.cat([nat.st[r][name][i] for r in range(P)])
            tol = bound(S[1][name][i])
            assert float((got - rf).abs().max() / rf.abs().max()) <= tol, ("slabs vs single domain without remove_divergence", name, i)
            ob = torch.from_numpy(B[1][name][i]).cuda()
            assert float((got - ob).abs().max() / ob.abs().max()) <= tol, ("slabs vs oracle without remove_divergence", name, i)
    diff = float((torch.cat([on.st[r]["q"][1] for r in range(P)]) - torch.cat([nat.st[r]["q"][1] for r in range(P)])).abs().max())
    assert diff > 1e-8, "the switch must change the wall-normal velocity"


def test_native_driver_refuses_thin_slabs_and_unbound_arrays(T):
    import ctypes
    from tlab_amd.lib import load, c_vp
    from tlab_amd.slab import NativeSlabDns, loopback_transport
    x, y, z = _grid(32, 16, 64)
    with pytest.raises(T.TlabError):      # kmax = 16: slab separators still couple (tlab_zslab_plan_create) -> the K-transposition scheme is the caller's
        NativeSlabDns("loopback", x, y, z, size=4, hyper_bc1_ext=REF_HYPER)
    x, y, z = _grid(32, 16, 128)
    L = load()
    g = [T.FdmPlan(x, True, True), T.FdmPlan(y, False, False), T.FdmPlan(z, True, True)]
    tr, _ = loopback_transport(2)
    h = c_vp(0)
    one = (ctypes.c_double * 1)(1.0)
    assert L.tlab_slab_dns_create(ctypes.byref(h), ctypes.byref(tr), g[0]._h, g[1]._h, g[2]._h, 32, 16, 128, 1, 1e-3, one, None) == 0
    assert L.tlab_slab_dns_substep(h, 1e-3, 1.0, 0) != 0 and b"bound" in L.tlab_last_error()
    assert L.tlab_slab_dns_destroy(h) == 0


def test_full_size_eight_native_slabs(T):
    """The strong-scaling case of the benchmark itself, 512^3 as 8 slabs of 64 planes (kx-pencils 33 + 7 x 32 in two halves), one RK3 step: the
    native driver against the Python driver to the bit, and against the single-domain driver within its own one-ulp scatter."""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    from tlab_amd.slab import NativeSlabDns
    n, P = 512, 8
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(8)
    Y = torch.arange(n, dtype=torch.float64, device="cuda").view(1, n, 1) / (n - 1)
    wall = torch.sin(np.pi * Y)
    fields = [((torch.rand(n, n, n, dtype=torch.float64, device="cuda", generator=gen) - 0.5) * wall).reshape(-1) for _ in range(4)]
    kw = dict(nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=REF_HYPER)
    one = Dns(x, y, x.copy(), **kw)
    for t, f in zip(one.q + one.s, fields):
        t.copy_(f)
    one.TIME_RUNGEKUTTA(1e-3)
    ref = [t.clone() for t in one.q + one.s]
    for t, f in zip(one.q + one.s, fields):
        r = torch.randint(-1, 2, f.shape, device="cuda", generator=gen)
        t.copy_(torch.where(r > 0, torch.nextafter(f, torch.full_like(f, 1e300)), torch.where(r < 0, torch.nextafter(f, torch.full_like(f, -1e300)), f)))
    one.TIME_RUNGEKUTTA(1e-3)
    scat = [float((t - rf).abs().max() / rf.abs().max()) for t, rf in zip(one.q + one.s, ref)]
    del one
    torch.cuda.empty_cache()
    out = {}
    for which in ("python", "native", "native-fused"):
        d = SlabDns(LoopbackComm(P), x, y, x.copy(), **kw) if which == "python" else \
            NativeSlabDns("loopback", x, y, x.copy(), size=P, fused_x=which == "native-fused", **kw)
        assert d.zmode == "halo" and d.stages == 2 and (which == "python" or d.fused_x == (which == "native-fused"))
        for i in range(3):
            d.scatter("q", i, fields[i])
        d.scatter("s", 0, fields[3])
        for k in range(3):
            d.substep_of_cycle(k, 1e-3)
        torch.cuda.synchronize()
        out[which] = [torch.cat([d.st[r][nm][ix] for r in range(P)]) for nm, ix in (("q", 0), ("q", 1), ("q", 2), ("s", 0))]
        del d
        torch.cuda.empty_cache()
    for i, rf in enumerate(ref):
        assert torch.equal(out["python"][i], out["native"][i]), i
        for which in ("native", "native-fused"):
            err = float((out[which][i] - rf).abs().max() / rf.abs().max())
            assert err <= bound(scat[i]), (which, i, err, scat[i])
