"""BASELINE.json configs[1], [3] and [4] AT SIZE on one MI355X (configs[2], the 512^3 box of the benchmark, is in test_gpu_rhs.py /
test_gpu_slab.py).  Where the numpy oracle cannot go (5e8 points) the checks are the size-independent properties of the path:
  * the fused driver (multi-field Burgers launches, operand / gradient / update fusions, chunked per-mode solver, own z-FFT) against the
    literal operator sequence of the reference (set_fusion(False)) -- two different kernel sets for the same equations;
  * the slab algorithm of 8 ranks (halo-partitioned z systems, kx-pencil Poisson) against the single domain;
  * the projected velocity is solenoidal in the interior;
each bounded by max(1e-12, 2 x the measured one-ulp scatter of the compared path itself) (tests/scatter.py)."""
import numpy as np
import pytest
from conftest import rel_err
from scatter import bound

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


def device_fields(nx, ny, nz, nfields, seed, y=None):
    """SURVEY.md 8d synthetic fields, generated on the device: smooth modes + 0.1 uniform noise, zero on the walls."""
    import torch
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    X = torch.arange(nx, dtype=torch.float64, device="cuda").view(1, 1, nx) / nx
    Y = (torch.arange(ny, dtype=torch.float64, device="cuda") / (ny - 1) if y is None else torch.from_numpy(np.ascontiguousarray(y)).cuda()).view(1, ny, 1)
    Z = torch.arange(nz, dtype=torch.float64, device="cuda").view(nz, 1, 1) / nz
    wall = torch.sin(np.pi * (Y - Y.min()) / (Y.max() - Y.min()))
    tp = 2 * np.pi
    out = []
    for k in range(nfields):
        sh = torch.sin(tp * (1 + k % 3) * X + k) * torch.cos(tp * (1 + k % 2) * Y) * torch.sin(tp * (1 + (k + 1) % 3) * Z + 0.5 * k)
        out.append(((sh + 0.1 * (2 * torch.rand(nz, ny, nx, dtype=torch.float64, device="cuda", generator=gen) - 1)) * wall).reshape(-1))
        del sh
    return out


def one_ulp_noise_(t, gen):
    """in place: every element moved by -1, 0 or +1 unit in the last place (device version of scatter.one_ulp_noise), in slices to bound temporaries"""
    import torch
    step = 1 << 26
    for a in range(0, t.numel(), step):
        v = t[a:a + step]
        r = torch.randint(-1, 2, v.shape, device="cuda", generator=gen)
        big = torch.full_like(v, 1e300)
        v.copy_(torch.where(r > 0, torch.nextafter(v, big), torch.where(r < 0, torch.nextafter(v, -big), v)))


def interior_divergence(T, d, margin):
    """max |div q| over the rows further than `margin` from the walls, relative to max|u| / h_x (FI_INVARIANT_P on the device)"""
    d.FI_INVARIANT_P(d.txc[0], d.txc[1])
    div = d.txc[0][: d.n].view(d.nz, d.ny, d.nx)[:, margin:-margin, :]
    scale = float(d.q[0].abs().max()) * d.nx
    return float(div.abs().max()) / scale


def test_configs1_burgers_and_partial_at_256_cubed_vs_oracle(T):
    """configs[1]: 256^3, 1 scalar, OPR_Partial + OPR_Burgers only -- at full size against the numpy oracle (16.8e6 points per call)."""
    import torch
    from oracle import tlab_oracle as O
    n = 256
    x = np.arange(n) / n
    y = 0.5 * (1 + np.tanh(2.0 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2.0))
    rng = np.random.default_rng(256)
    s = rng.uniform(-1, 1, n ** 3)
    u = rng.uniform(-1, 1, n ** 3)
    ds, du = torch.from_numpy(s).cuda(), torch.from_numpy(u).cuda()
    res, tmp = torch.empty_like(ds), torch.empty_like(ds)
    burg = (T.OPR_Burgers_X, T.OPR_Burgers_Y, T.OPR_Burgers_Z)
    part = (T.OPR_Partial_X, T.OPR_Partial_Y, T.OPR_Partial_Z)
    for d, (nodes, per, uni) in {1: (x, True, True), 2: (y, False, False), 3: (x, True, True)}.items():
        gp, op = T.FdmPlan(nodes, per, uni), O.FdmPlan(nodes, per, uni)
        burg[d - 1](T.OPR_B_U_IN, 1.0 / 5000.0, n, n, n, 0, gp, ds, du, res, tmp)
        ref = O.opr_burgers(d, n, n, n, 0, op, 1.0 / 5000.0, s, u)[0]
        assert rel_err(res.cpu().numpy(), ref) <= 1e-12, ("burgers", d)
        burg[d - 1](T.OPR_B_SELF, 1.0 / 5000.0, n, n, n, 0, gp, ds, ds, res, tmp)
        ref = O.opr_burgers(d, n, n, n, 0, op, 1.0 / 5000.0, s, s)[0]
        assert rel_err(res.cpu().numpy(), ref) <= 1e-12, ("burgers self", d)
        part[d - 1](T.OPR_P2_P1, n, n, n, 0, gp, ds, res, tmp)
        r2, r1 = O.opr_partial(d, O.OPR_P2_P1, n, n, n, 0, op, s)
        assert rel_err(res.cpu().numpy(), r2) <= 1e-12 and rel_err(tmp.cpu().numpy(), r1) <= 1e-12, ("partial", d)


def test_configs4_rank_share_fused_vs_literal_and_divergence(T):
    """configs[4]: 2048 x 1024 x 2048 with 3 scalars on a stretched y mesh over 8 GPUs -> one rank's share, 2048 x 1024 x 256 (5.4e8 points,
    ~125 GB of the 288), one full RK3 step: x lines of 2048 (32 rows per lane), stretched y lines of 1024 (16-line tiles, Jacobian correction),
    two Burgers launches per direction (4 + 2 fields), Poisson with 1025 x 256 modes of 1024 rows.  Then the same box as FOUR z-slabs of 64 planes
    through the native slab driver (all ranks on this GPU, exchanges as device copies): the composition of that shape -- 2048-point x lines and 1024-point
    y lines inside slabs, six transported fields in two launches per direction and phase, kx-pencils of 256-257 modes in two halves, packed x-transforms,
    v and the three scalars finished by their last kernels -- against the single domain within the scatter bound (the full 2048-plane box of the config
    would need 8 x this memory)."""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.slab import NativeSlabDns
    nx, ny, nz, ns = 2048, 1024, 256, 3
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz * 0.25
    y = 0.5 * (1 + np.tanh(2.0 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(2.0))          # SURVEY 8d's stretched mesh
    kw = dict(nscal=ns, visc=1.0 / 5000.0, schmidt=(1.0, 0.7, 2.0), yuniform=False, hyper_bc1_ext=0.0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(4)
    results = {}
    for mode in ("fused", "fused+ulp", "literal"):
        fields = device_fields(nx, ny, nz, 3 + ns, 2048, y)
        if mode == "fused+ulp":
            for f in fields:
                one_ulp_noise_(f, gen)
        d = Dns(x, y, z, **kw)
        d.set_fusion(mode != "literal")
        for t, f in zip(d.q + d.s, fields):
            t.copy_(f)
        del fields
        d.TIME_RUNGEKUTTA(2e-4)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(t).all()) for t in d.q + d.s)
        if mode == "fused":
            div = interior_divergence(T, d, 60)
            assert div <= 1e-10, div
            results["fused"] = [t.clone() for t in d.q + d.s]
        else:
            results[mode] = [float((t - r).abs().max() / r.abs().max()) for t, r in zip(d.q + d.s, results["fused"])]
        del d
        torch.cuda.empty_cache()
    print("configs[4] share: one-ulp scatter", ["%.1e" % v for v in results["fused+ulp"]], "fused vs literal", ["%.1e" % v for v in results["literal"]])
    for i, (err, sc) in enumerate(zip(results["literal"], results["fused+ulp"])):
        assert err <= bound(sc), (i, err, sc)
    # ---- four z-slabs ----
    P = 4
    d = NativeSlabDns("loopback", x, y, z, size=P, **kw)
    assert d.kmax == nz // P and d.stages == 2 and d.fused_x
    fields = device_fields(nx, ny, nz, 3 + ns, 2048, y)
    for i in range(3 + ns):
        d.scatter("q" if i < 3 else "s", i if i < 3 else i - 3, fields[i])
        fields[i] = None
    del fields
    torch.cuda.empty_cache()
    for k in range(3):
        d.substep_of_cycle(k, 2e-4)
    torch.cuda.synchronize()
    errs = []
    for i, rf in enumerate(results["fused"]):
        name, ix = ("q", i) if i < 3 else ("s", i - 3)
        scale = float(rf.abs().max())
        err = max(float((d.st[r][name][ix] - rf[r * d.n:(r + 1) * d.n]).abs().max()) for r in range(P)) / scale
        errs.append(err)
    print("configs[4] share as 4 slabs vs single domain", ["%.1e" % v for v in errs])
    d.close()
    for i, (err, sc) in enumerate(zip(errs, results["fused+ulp"])):
        assert err <= bound(sc), ("4 slabs", i, err, sc)


def test_configs3_eight_loopback_ranks_equal_single_domain(T):
    """configs[3]: 1024 x 512 x 1024 over 8 ranks (z-slabs of 128 planes) through the native C++ slab driver, every rank's work executed on this one
    GPU (loopback transport: only the exchanges are copies), against the single-domain driver after one RK3 step; interior divergence of the slab
    result.  (The Python statement of the driver runs the same case bit-identically at smaller sizes: tests/test_gpu_slab_native.py.)"""
    import torch
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, LoopbackComm
    nx, ny, nz, P = 1024, 512, 1024, 8
    x = np.arange(nx) / nx * 2.0
    y = np.arange(ny) / (ny - 1.0)
    z = np.arange(nz) / nz * 2.0
    kw = dict(nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=0.0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    fields = device_fields(nx, ny, nz, 4, 1024)
    one = Dns(x, y, z, **kw)
    for t, f in zip(one.q + one.s, fields):
        t.copy_(f)
    one.TIME_RUNGEKUTTA(5e-4)
    ref = [t.clone() for t in one.q + one.s]
    div = interior_divergence(T, one, 40)
    assert div <= 1e-10, div
    for t, f in zip(one.q + one.s, fields):          # conditioning: the same step from fields one ulp of white noise away
        t.copy_(f)
        one_ulp_noise_(t, gen)
    one.TIME_RUNGEKUTTA(5e-4)
    scat = [float((t - r).abs().max() / r.abs().max()) for t, r in zip(one.q + one.s, ref)]
    del one
    torch.cuda.empty_cache()
    # the NATIVE driver (tlab_slab_dns_*, csrc/slab.cpp: the code a Fortran / MPI host runs) at the size of configs[3]
    from tlab_amd.slab import NativeSlabDns
    slab = NativeSlabDns("loopback", x, y, z, size=P, **kw)
    assert slab.zmode == "halo" and slab.kmax == 128 and slab.stages == 2
    for i in range(3):
        slab.scatter("q", i, fields[i])
    slab.scatter("s", 0, fields[3])
    del fields
    for k in range(3):
        slab.substep_of_cycle(k, 5e-4)
    torch.cuda.synchronize()
    errs = []
    for i, rf in enumerate(ref):
        name, idx = ("q", i) if i < 3 else ("s", 0)
        got = torch.cat([slab.st[r][name][idx] for r in range(P)])
        assert bool(torch.isfinite(got).all())
        errs.append(float((got - rf).abs().max() / rf.abs().max()))
        del got
    slab.close()
    print("configs[3] loopback-8 (native driver): one-ulp scatter", ["%.1e" % v for v in scat], "slabs vs single domain", ["%.1e" % v for v in errs])
    for i, (e, sc) in enumerate(zip(errs, scat)):
        assert e <= bound(sc), (i, e, sc)
    # the same box "x/z-decomposed" as BASELINE configs[3] words it: 2 x 4 blocks of 512 x 512 x 256 (tlab_amd/pencil.py: I-transpositions in
    # the x communicators, K-transpositions in the z communicators, Poisson on the 1 x 8 slabs the I-transposition leaves)
    fields = device_fields(nx, ny, nz, 4, 1024)
    del slab
    torch.cuda.empty_cache()
    from tlab_amd.pencil import PencilDns, loopback_comms
    pen = PencilDns(loopback_comms(2, 4), 2, 4, x, y, z, **kw)
    assert (pen.imax, pen.kmax, pen.kmax2) == (512, 256, 128)
    for i in range(3):
        pen.scatter("q", i, fields[i])
    pen.scatter("s", 0, fields[3])
    del fields
    for k in range(3):
        pen.substep_of_cycle(k, 5e-4)
    torch.cuda.synchronize()
    errs = []
    for i, rf in enumerate(ref):
        name, idx = ("q", i) if i < 3 else ("s", 0)
        got = torch.empty(nz, ny, nx, dtype=torch.float64, device="cuda")
        for r, t in pen.gather_local(name, idx).items():
            pi, pk = pen.pro(r)
            got[pk * pen.kmax:(pk + 1) * pen.kmax, :, pi * pen.imax:(pi + 1) * pen.imax] = t.view(pen.kmax, ny, pen.imax)
        got = got.reshape(-1)
        assert bool(torch.isfinite(got).all())
        errs.append(float((got - rf).abs().max() / rf.abs().max()))
        del got
    print("configs[3] loopback 2 x 4: pencils vs single domain", ["%.1e" % v for v in errs])
    for i, (e, sc) in enumerate(zip(errs, scat)):
        assert e <= bound(sc), (i, e, sc)
