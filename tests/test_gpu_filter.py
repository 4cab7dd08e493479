"""[Dealiasing]: the 1-D filters of OPR_FILTER_1D on the device (tlab_amd/csrc/filter.hip) against vectors the reference's own filter modules
produced (tests/golden/filters.npz) and against the oracle in the three directions; the dealiasing branch of OPR_Burgers and a substep with it."""
import re
import numpy as np
import pytest
from conftest import rel_err
from test_oracle_filter import G, filter_of
from scatter import substep_scatter, bound

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


def device_filter(T, key):
    n, t, per, b0, b1 = (int(v) for v in re.match(r"n(\d+)_t(\d+)_p(\d)_b(\d)(\d)", key).groups())
    c = G[key + "_coeffs"]
    return T.Filter(t, n, bool(per), c if c.shape[1] else None, b0, b1), n


@pytest.mark.parametrize("key", [str(k) for k in G["cases"]])
def test_filters_match_the_reference_vectors(T, key):
    """every case of the fixture as y lines (the lines of the fixture side by side as an (nx, n, 1) box) and as x lines (n, nlines, 1)"""
    import torch
    f, n = device_filter(T, key)
    u = G["n%d_u" % n]                      # (n, nlines)
    want = G[key + "_res"]
    nl = u.shape[1]
    du = torch.from_numpy(np.ascontiguousarray(u)).cuda().reshape(-1)          # y lines: index = i + nl * j
    res = torch.full_like(du, float("nan"))
    T.OPR_FILTER_1D(2, f, nl, n, 1, du, res)
    assert rel_err(res.cpu().numpy().reshape(n, nl), want) <= 1e-14, key
    dx = torch.from_numpy(np.ascontiguousarray(u.T)).cuda().reshape(-1)        # x lines: index = i + n * line
    resx = torch.full_like(dx, float("nan"))
    T.OPR_FILTER_1D(1, f, n, nl, 1, dx, resx)
    assert rel_err(resx.cpu().numpy().reshape(nl, n).T, want) <= 1e-14, key
    dz = torch.from_numpy(np.ascontiguousarray(u)).cuda().reshape(-1)          # z lines of a (nl, 1, n) box
    resz = torch.full_like(dz, float("nan"))
    T.OPR_FILTER_1D(3, f, nl, 1, n, dz, resz)
    assert rel_err(resz.cpu().numpy().reshape(n, nl), want) <= 1e-14, key


def test_unsupported_filters_are_refused(T):
    with pytest.raises(T.TlabError):
        T.Filter(8, 32, True)                        # tophat
    with pytest.raises(T.TlabError):
        T.Filter(1, 32, True)                        # compact without its table


def test_dealiased_burgers_and_substep_vs_oracle(T):
    """[Dealiasing] Type = compactcutoff-like setup: compact cutoff in the periodic directions, the compact filter in y
    (what FILTER_READBLOCK makes of 'compactcutoff', opr_filter.f90:136-139)."""
    import torch
    from oracle import tlab_oracle as O
    from oracle.tlab_oracle_rhs import DnsOracle
    from tlab_amd.dns import Dns
    from test_gpu_rhs import grids, init_fields
    nx, ny, nz = 64, 64, 64
    x, y, z = grids(nx, ny, nz, True)
    assert np.array_equal(x, G["n64_x"]) and np.array_equal(y, G["n64_y"])       # the tables of the fixture belong to these nodes
    keys = {1: "n64_t9_p1_b00", 2: "n64_t1_p0_b11", 3: "n64_t9_p1_b00"}          # (the periodic cutoff filter does not depend on the nodes)
    dev = {d: device_filter(T, k)[0] for d, k in keys.items()}
    orc = {d: filter_of(k)[0] for d, k in keys.items()}
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 7)
    visc, sc = 1.0 / 500.0, (0.7,)
    try:
        for d in (1, 2, 3):
            T.set_dealiasing(d, dev[d])
        # operator level
        s_, v_ = torch.from_numpy(s0[0]).cuda(), torch.from_numpy(q0[1]).cuda()
        res, tmp = torch.empty_like(s_), torch.empty_like(s_)
        for d, (nodes, per) in {1: (x, True), 2: (y, False), 3: (z, True)}.items():
            gp, go = T.FdmPlan(nodes, per, per, hyper_bc1_ext=REF_HYPER), O.FdmPlan(nodes, per, per)
            (T.OPR_Burgers_X, T.OPR_Burgers_Y, T.OPR_Burgers_Z)[d - 1](T.OPR_B_U_IN, visc, nx, ny, nz, 0, gp, s_, v_, res, tmp)
            ref = O.opr_burgers(d, nx, ny, nz, 0, go, visc, s0[0], q0[1], dealiasing=orc[d])[0]
            assert rel_err(res.cpu().numpy(), ref) <= 1e-12, d
        # two substeps
        dn = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
        for i in range(3):
            dn.q[i].copy_(torch.from_numpy(q0[i]))
        dn.s[0].copy_(torch.from_numpy(s0[0]))
        sched = [(2e-3 * dn.kdt[k], dn.kco[k], True) for k in range(2)]
        B, S = substep_scatter(lambda: DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, dealiasing=[orc[1], orc[2], orc[3]]),
                               q0, s0, sched, nsamples=2)
        for k, (dte, kco, scale) in enumerate(sched):
            dn.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
            for name in ("q", "hq", "s", "hs"):
                for i, (b, scat) in enumerate(zip(B[k][name], S[k][name])):
                    e = rel_err(getattr(dn, name)[i].cpu().numpy(), b)
                    assert e <= bound(scat), (k, name, i, e, scat)
        plain = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        for i in range(3):
            plain.q[i] = q0[i].copy()
        plain.s[0] = s0[0].copy()
        plain.time_substep(*sched[0])
        assert rel_err(B[0]["hq"][0], plain.hq[0]) > 1e-6            # (the filters do something)
    finally:
        for d in (1, 2, 3):
            T.set_dealiasing(d, None)


def test_pressure_filter_substeps_vs_oracle(T):
    """[PressureFilter] Type = compact, ActiveY only, BcsJmin = BcsJmax = zero (examples/Case92-93): p and dp/dy pass OPR_FILTER after the Poisson
    solve (rhs_global_incompressible_1.f90:286-290)."""
    import torch
    from oracle.tlab_oracle_rhs import DnsOracle
    from tlab_amd.dns import Dns
    from test_gpu_rhs import grids, init_fields
    nx, ny, nz = 64, 64, 32
    x, y, z = grids(nx, ny, nz, True)
    assert np.array_equal(y, G["n64_y"])
    fdev, _ = device_filter(T, "n64_t1_p0_b66")
    forc, _ = filter_of("n64_t1_p0_b66")
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 21)
    visc, sc = 1.0 / 500.0, (0.7,)
    dn = Dns(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False, hyper_bc1_ext=REF_HYPER)
    dn.set_pressure_filter(None, fdev, None)
    for i in range(3):
        dn.q[i].copy_(torch.from_numpy(q0[i]))
    dn.s[0].copy_(torch.from_numpy(s0[0]))
    sched = [(2e-3 * dn.kdt[k], dn.kco[k], True) for k in range(2)]

    def make_oracle():
        o = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
        o.pressure_filter = [None, forc, None]
        return o
    B, S = substep_scatter(make_oracle, q0, s0, sched, nsamples=2)
    for k, (dte, kco, scale) in enumerate(sched):
        dn.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dte, kco, scale)
        for name in ("q", "hq", "s", "hs"):
            for i, (b, scat) in enumerate(zip(B[k][name], S[k][name])):
                e = rel_err(getattr(dn, name)[i].cpu().numpy(), b)
                assert e <= bound(scat), (k, name, i, e, scat)
    plain = DnsOracle(x, y, z, nscal=1, visc=visc, schmidt=sc, yuniform=False)
    for i in range(3):
        plain.q[i] = q0[i].copy()
    plain.s[0] = s0[0].copy()
    plain.time_substep(*sched[0])
    assert rel_err(B[0]["q"][1], plain.q[1]) > 1e-9
