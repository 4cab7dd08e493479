"""x/z pencil decomposition (ims_npro_i x ims_npro_k, tlab_amd/pencil.py) against the single domain on ONE GPU: all ranks are simulated
in-process (loopback communicators), so the complete decomposed algorithm -- I- and K-transpositions inside the x / z communicators, the
transposed-velocity reuse, the Poisson solver on the 1 x (npro_i npro_k) slabs the I-transposition leaves behind, per-rank kx ranges -- runs
for real; only the transport of the blocks is replaced by copies."""
import numpy as np
import pytest
from scatter import bound, substep_scatter, ref_of
import cases as C

pytestmark = pytest.mark.gpu
REF_HYPER = 0.1


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


def grids(nx, ny, nz):
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * np.pi
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    return x, y, z


def gather(D, name, idx, nx, ny, nz):
    import torch
    out = torch.empty(nz, ny, nx, dtype=torch.float64)
    for r, t in D.gather_local(name, idx).items():
        pi, pk = D.pro(r)
        out[pk * D.kmax:(pk + 1) * D.kmax, :, pi * D.imax:(pi + 1) * D.imax] = t.cpu().view(D.kmax, D.ny, D.imax)
    return out.reshape(-1).numpy()


@pytest.mark.parametrize("npi,npk,nx,ny,nz,bcs", [(2, 2, 32, 24, 16, "noslip"), (2, 4, 32, 16, 32, "noslip"), (4, 2, 64, 16, 16, "freeslip"),
                                                   (2, 1, 32, 24, 8, "noslip"), (8, 1, 64, 8, 16, "noslip"), (1, 4, 32, 16, 16, "noslip")])
def test_pencil_substeps_match_single_domain(T, npi, npk, nx, ny, nz, bcs):
    import torch
    from tlab_amd.dns import Dns, velocity_bcs
    from tlab_amd.pencil import PencilDns, loopback_comms
    from oracle.tlab_oracle_rhs import DnsOracle
    case = C.pencil(npi, npk, nx, ny, nz, bcs)
    x, y, z, visc, sc = (case[k] for k in ("x", "y", "z", "visc", "sc"))
    fields = case["q0"] + case["s0"]
    one = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    D = PencilDns(loopback_comms(npi, npk), npi, npk, x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    if bcs == "freeslip":
        one.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
        D.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
    for i in range(3):
        t = torch.from_numpy(fields[i]).cuda()
        one.q[i].copy_(t); D.scatter("q", i, t)
    t = torch.from_numpy(fields[3]).cuda()
    one.s[0].copy_(t); D.scatter("s", 0, t)
    dtime = 2e-3
    for k in range(2):
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], one.kco[k], True)
        D.substep_of_cycle(k, dtime)
    torch.cuda.synchronize()

    # the bound: max(1e-12, 2 x the ORACLE's own scatter under one ulp of input noise) (tests/scatter.py), against the single domain and the oracle
    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        if bcs == "freeslip":
            o.flow_jmin = o.flow_jmax = velocity_bcs("freeslip"); o.scal_jmin, o.scal_jmax = [4], [3]
        return o
    assert [(dtime * one.kdt[k], one.kco[k], True) for k in range(2)] == case["sched"]
    B, S = substep_scatter(make_oracle, fields[:3], fields[3:], case["sched"], nsamples=2)
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s), ("hs", one.hs)):
        for i, rf in enumerate(ref):
            got = gather(D, name, i, nx, ny, nz)
            tol = bound(S[1][name][i], ref=ref_of(case["key"], 1, name, i))
            rf = rf.cpu().numpy()
            err = float(np.abs(got - rf).max() / np.abs(rf).max())
            assert err <= tol, ("pencils vs single domain", name, i, err, tol)
            err = float(np.abs(got - B[1][name][i]).max() / np.abs(B[1][name][i]).max())
            assert err <= tol, ("pencils vs oracle", name, i, err, tol)


def test_pencil_layout_is_refused_where_the_reference_refuses(T):
    from tlab_amd.pencil import PencilDns, loopback_comms
    x, y, z = grids(32, 16, 12)
    with pytest.raises(T.TlabError):
        PencilDns(loopback_comms(4, 2), 4, 2, x, y, z)          # kmax = 6 is not a multiple of npro_i = 4


@pytest.mark.parametrize("npi,npk,nx,ny,nz,bcs,ns", [(2, 2, 32, 24, 16, "noslip", 1), (2, 4, 32, 16, 32, "freeslip", 2), (4, 2, 64, 16, 16, "noslip", 1),
                                                      (8, 1, 64, 8, 16, "noslip", 0), (1, 4, 32, 16, 16, "freeslip", 1), (4, 4, 64, 16, 32, "noslip", 1)])
def test_native_pencil_driver_is_bit_identical_to_the_python_driver(T, npi, npk, nx, ny, nz, bcs, ns):
    """tlab_pencil_dns_* (tlab_amd/csrc/pencil.cpp: what a Fortran / MPI host calls) against tlab_amd/pencil.py::PencilDns, both on loopback ranks:
    the same library calls on the same numbers in the same order, the transpositions and the three all-to-alls as native index work -> every field
    of every rank equal to the bit after a full Runge-Kutta step (three substeps incl. the tendency scaling and the step-start zeroing)."""
    import torch
    from tlab_amd.pencil import PencilDns, NativePencilDns, loopback_comms
    x, y, z = grids(nx, ny, nz)
    rng = np.random.default_rng(100 * npi + npk)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    fields = [((np.sin(X + k) * np.cos(2 * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(3 + ns)]
    kw = dict(nscal=ns, visc=1.0 / 300.0, schmidt=(0.7, 1.3)[:ns], yuniform=False, hyper_bc1_ext=REF_HYPER)
    py = PencilDns(loopback_comms(npi, npk), npi, npk, x, y, z, **kw)
    nat = NativePencilDns("loopback", npi, npk, x, y, z, **kw)
    assert (nat.imax, nat.kmax, nat.kmax2, nat.isize_txc) == (py.imax, py.kmax, py.kmax2, py.isize_txc)
    if bcs == "freeslip":
        py.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
        nat.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
    for d in (py, nat):
        for i in range(3 + ns):
            d.scatter("q" if i < 3 else "s", i if i < 3 else i - 3, torch.from_numpy(fields[i]).cuda())
    for k in range(4):
        py.substep_of_cycle(k, 2e-3)
        nat.substep_of_cycle(k, 2e-3)
    torch.cuda.synchronize()
    for name in ("q", "s", "hq", "hs"):
        for r in range(npi * npk):
            for a, b in zip(py.st[r][name], nat.st[r][name]):
                assert float(b.abs().max()) > 0.0 and torch.equal(a, b), (name, r)
    nat.close()


def test_native_pencil_driver_with_redrawn_arrays_gives_the_same_run(T):
    """NativePencilDns.redraw_arrays (tlab_amd/placement.py; what bench.py --decomp does before its timed region): other allocations, same run."""
    import torch
    from tlab_amd.pencil import NativePencilDns
    npi, npk, nx, ny, nz = 2, 2, 32, 24, 16
    x, y, z = grids(nx, ny, nz)
    rng = np.random.default_rng(17)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    fields = [((np.sin(X + k) * np.cos(2 * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(4)]
    kw = dict(nscal=1, visc=1.0 / 300.0, schmidt=(0.7,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    a, b = NativePencilDns("loopback", npi, npk, x, y, z, **kw), NativePencilDns("loopback", npi, npk, x, y, z, **kw)
    for d in (a, b):
        for i in range(4):
            d.scatter("q" if i < 3 else "s", i if i < 3 else i - 3, torch.from_numpy(fields[i]).cuda())
    a.substep_of_cycle(0, 2e-3); b.substep_of_cycle(0, 2e-3)
    b.redraw_arrays(pool=24, seed=2)
    for k in range(1, 4):
        a.substep_of_cycle(k, 2e-3); b.substep_of_cycle(k, 2e-3)
    torch.cuda.synchronize()
    for name in ("q", "s", "hq", "hs"):
        for r in range(npi * npk):
            for u, v in zip(a.st[r][name], b.st[r][name]):
                assert float(v.abs().max()) > 0.0 and torch.equal(u, v), (name, r)
    a.close(); b.close()


@pytest.mark.parametrize("npi,npk,ns", [(2, 2, 1), (2, 4, 2), (4, 1, 0), (1, 4, 3)])
def test_native_pencil_driver_starts_its_transpositions_ahead_of_independent_launches(T, npi, npk, ns, monkeypatch):
    """The overlapped schedule of tlab_pencil_dns_rhs (csrc/pencil.cpp::rhs_overlapped; reference: rhs_global_incompressible_nbc.f90:135-382): between EVERY
    exchange start and its wait stands at least one launch that does not depend on it -- the operator of the previous field, a local y operator, a sum,
    or the start / packing of the next exchange's payload -- read off the driver's own record of the order it issued things in
    (tlab_pencil_dns_trace).  And the schedule changes no bit: the same fields as the literal sequence (TLAB_PENCIL_OVERLAP=0) after a full RK step."""
    import ctypes
    import torch
    from tlab_amd.lib import load
    from tlab_amd.pencil import NativePencilDns
    nx, ny, nz = 64, 16, 32
    x, y, z = grids(nx, ny, nz)
    rng = np.random.default_rng(7 * npi + npk + ns)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    fields = [((np.sin(X + k) * np.cos(2 * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(3 + ns)]
    kw = dict(nscal=ns, visc=1.0 / 300.0, schmidt=(0.7, 1.3, 2.0)[:ns], yuniform=False, hyper_bc1_ext=REF_HYPER)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TLAB_PENCIL_OVERLAP", mode)
        d = NativePencilDns("loopback", npi, npk, x, y, z, **kw)
        for i in range(3 + ns):
            d.scatter("q" if i < 3 else "s", i if i < 3 else i - 3, torch.from_numpy(fields[i]).cuda())
        L = load()
        L.tlab_pencil_dns_trace(d._h, 1, None, 0)
        for k in range(3):
            d.substep_of_cycle(k, 2e-3)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        L.tlab_pencil_dns_trace(d._h, 0, buf, len(buf))
        out[mode] = ([t.clone() for r in range(npi * npk) for name in ("q", "s", "hq", "hs") for t in d.st[r][name]], buf.value.decode().splitlines())
        d.close()
    for a, b in zip(out["1"][0], out["0"][0]):
        assert torch.equal(a, b)
    events = out["1"][1]
    assert out["0"][1] == []                                     # (the literal sequence records nothing: every start is followed by its wait)
    starts = [i for i, e in enumerate(events) if e.startswith("start ")]
    # transposed operators of ONE RHS (the record is per call): 3 + ns Burgers terms + 2 first derivatives per split direction, a forward and a backward exchange each
    n_exchanges = (3 + ns + 2) * 2 * ((npi > 1) + (npk > 1))
    assert len(starts) == n_exchanges, (len(starts), n_exchanges, events[:20])
    for i in starts:
        kind, which = events[i].split(" ", 1)[1].rsplit(" ", 1)
        j = events.index("wait %s %s" % (kind, which), i)       # its wait (per RHS call the pairs are unique up to the pipeline restarts: first match after i)
        between = events[i + 1:j]
        idle_ok = "nothing left to overlap" in between
        assert any(e.startswith("launch") for e in between) or idle_ok, (events[i], between)
    # the driver says so itself where a pipeline has run out of independent work before its LAST backward transposition: at most once per pipeline (three per
    # RHS: the Burgers terms, the divergence, the pressure gradient), never in the Burgers pipeline
    assert sum(1 for e in events if e == "nothing left to overlap") <= 2, events


def test_native_pencil_driver_refusals(T):
    from tlab_amd.pencil import NativePencilDns
    x, y, z = grids(32, 16, 12)
    with pytest.raises(T.TlabError, match="npro_i must divide kmax"):
        NativePencilDns("loopback", 4, 2, x, y, z)
    x, y, z = grids(32, 16, 16)
    nat = NativePencilDns("loopback", 2, 2, x, y, z)
    nat.close()
    import ctypes
    from tlab_amd.lib import load, c_vp
    from tlab_amd.pencil import PencilTransport
    L = load()
    g = [T.FdmPlan(x, True, True), T.FdmPlan(y, False, False), T.FdmPlan(z, True, True)]
    tr = PencilTransport()
    assert L.tlab_pencil_transport_loopback(ctypes.byref(tr), 2, 2) == 0
    h = c_vp(0)
    one = (ctypes.c_double * 1)(1.0)
    assert L.tlab_pencil_dns_create(ctypes.byref(h), ctypes.byref(tr), g[0]._h, g[1]._h, g[2]._h, 32, 16, 16, 1, 1e-3, one) == 0
    assert L.tlab_pencil_dns_substep(h, 1e-3, 1.0, 0) != 0 and b"bound" in L.tlab_last_error()
    assert L.tlab_pencil_dns_destroy(h) == 0
