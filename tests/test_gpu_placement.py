"""tlab_dns_place_arrays (include/tlab_amd.h): the search for the allocations on which the substep runs fastest changes WHERE q, s, hq, hs, txc live and
nothing else -- a run after it is the run before it, bit for bit."""
import numpy as np
import pytest

import cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    T.init(0)
    return T


def _dns(nx, ny, nz, seed, nscal=1):
    import torch
    from tlab_amd.dns import Dns, RKM_EXP3
    x, y, z = C.grids(nx, ny, nz, True)
    d = Dns(x, y, z, nscal=nscal, visc=1.0 / 500.0, schmidt=(0.7, 1.2)[:nscal], yuniform=False, rkm_mode=RKM_EXP3, hyper_bc1_ext=0.1)
    q0, s0 = C.init_fields(nx, ny, nz, x, y, z, seed)
    s0 = (s0 + [0.5 * s0[0]])[:nscal]
    for t, a in zip(d.q + d.s, q0 + s0):
        t.copy_(torch.from_numpy(a))
    return d


@pytest.mark.parametrize("nscal", [1, 0, 2])
def test_placed_arrays_give_the_same_run_bit_for_bit(T, nscal):
    import torch
    nx, ny, nz = 256, 64, 32
    a, b = _dns(nx, ny, nz, 5, nscal), _dns(nx, ny, nz, 5, nscal)
    nroles = 2 * (3 + nscal) + 9
    rep = b.place_arrays(pool=24, random_trials=3, dtime=1e-3, seed=3)
    after = [t.data_ptr() for t in b.q + b.s + b.hq + b.hs + b.txc]
    assert len(set(after)) == nroles                                                # distinct arrays out of the pool
    assert rep["pool"] == 24 and rep["trials"] == 1 + 3 + nroles
    assert 0.0 < rep["ms_best"] <= rep["ms_first"] and rep["ms_median"] <= rep["ms_worst"]      # (first / best: the repeats at the end of the search)
    for t, u in zip(a.q + a.s, b.q + b.s):                                          # the fields came along
        assert torch.equal(t, u)
    for d in (a, b):
        for _ in range(2):
            d.TIME_RUNGEKUTTA(1e-3)
    torch.cuda.synchronize()
    for t, u in zip(a.q + a.s + a.hq + a.hs, b.q + b.s + b.hq + b.hs):
        assert torch.equal(t, u)
    assert all(bool(torch.isfinite(t).all()) for t in b.q + b.s)


def test_place_arrays_refuses_a_pool_that_is_too_small(T):
    import ctypes
    import torch
    from tlab_amd.lib import load, c_vp, TlabError, check
    d = _dns(256, 64, 32, 6)
    m = d.isize_txc_field
    pool = [torch.zeros(m, dtype=torch.float64, device="cuda") for _ in range(16)]      # 17 roles
    parr = (c_vp * 16)(*[t.data_ptr() for t in pool])
    assign = (ctypes.c_int * 17)()
    with pytest.raises(TlabError):
        check(load().tlab_dns_place_arrays(d._h, 16, parr, None, 1e-3, 2, 0, assign, None), "tlab_dns_place_arrays")
    parr2 = (c_vp * 17)(*([t.data_ptr() for t in pool] + [pool[0].data_ptr()]))        # the same array twice
    with pytest.raises(TlabError):
        check(load().tlab_dns_place_arrays(d._h, 17, parr2, None, 1e-3, 2, 0, assign, None), "tlab_dns_place_arrays")


def test_bench_line_of_one_gpu_carries_the_placement_report_and_times_the_dominant_kernel_only(T):
    """bench.py's default single-GPU line at a small box: the placement search ran before the timed region (its report is in the line), the roofline
    object comes from events around the dominant kernel only -- a few launches per timed substep -- and the table of all kernels from its own pass."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--n", "256", "--steps", "6", "--warmup", "3", "--cpu-sample", "0", "--no-freeslip-leg", "--no-fortran-host", "--placement-trials", "4",
           "--placement-pool", "24"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    pl = rec["placement"]
    assert pl["pool"] == 24 and pl["trials"] == 1 + 4 + 17 and 0 < pl["ms_best"] <= pl["ms_first"]
    rf = rec["roofline"]
    assert rf["bound"] == "hbm" and rf["launches"] in (6, 12, 18) and 0 < rf["frac"] < 1      # one kernel name: 1 .. 3 launches per substep
    names = [k["kernel"] for k in rec["kernels"]]
    assert rf["kernel"] in names and len(names) > 8 and "after the timed region" in rec["kernels_from"]
    assert rec.get("substep_traffic") is None or rec["substep_traffic"]["consistent"] is not False      # (512^3 only: a substep moves at least its dominant kernel's bytes)
    assert rec["config"]["fields_finite"] is True and rec["steps"] == 6


@pytest.mark.parametrize("ns", [1, 0])
def test_place_blocks_for_a_host_with_two_dimensional_arrays(T, ns):
    """tlab_dns_place_blocks: the host's layout -- q(n, 3), s(n, ns), hq, hs, txc(m, 9) as ONE allocation each -- with three candidates per block:
    the choice is one candidate per block, the begin_step flag of the driver survives the trial substeps, and a step on the chosen blocks equals the
    step on separately allocated arrays to the bit (where the arrays live changes nothing but the time)."""
    import ctypes
    import torch
    from tlab_amd.lib import load, c_vp, check, TlabError
    nx, ny, nz = 256, 64, 32
    ref = _dns(nx, ny, nz, 7, ns)
    d = _dns(nx, ny, nz, 7, ns)
    n, m = d.n, d.isize_txc_field
    ncand = 3
    blocks = {"q": [torch.zeros(3 * n, dtype=torch.float64, device="cuda") for _ in range(ncand)], "s": [torch.zeros(max(ns, 1) * n, dtype=torch.float64, device="cuda") for _ in range(ncand)],
              "hq": [torch.zeros(3 * n, dtype=torch.float64, device="cuda") for _ in range(ncand)], "hs": [torch.zeros(max(ns, 1) * n, dtype=torch.float64, device="cuda") for _ in range(ncand)],
              "txc": [torch.zeros(9 * m, dtype=torch.float64, device="cuda") for _ in range(ncand)]}
    arr = {k: (c_vp * ncand)(*[t.data_ptr() for t in v]) for k, v in blocks.items()}
    if ns == 0:      # a host without scalars hands over arrays of null pointers for s, hs (TLab_AMD_Place_Arrays): they are not looked at
        arr["s"] = (c_vp * ncand)()
        arr["hs"] = (c_vp * ncand)()
    choice = (ctypes.c_int * 5)()
    rep = (ctypes.c_double * 5)()
    d.begin_step()
    check(load().tlab_dns_place_blocks(d._h, ncand, arr["q"], arr["s"], arr["hq"], arr["hs"], arr["txc"], m, 1e-3, 4, 1, choice, rep), "tlab_dns_place_blocks")
    c = list(choice)
    assert all(0 <= v < ncand for v in c) and int(rep[4]) >= 1 + 4 + (5 if ns else 3) * (ncand - 1) and 0 < rep[1] <= rep[0]
    Q, S, HQ, HS, X = (blocks[k][c[i]] for i, k in enumerate(("q", "s", "hq", "hs", "txc")))
    d.q, d.s = [Q[i * n:(i + 1) * n] for i in range(3)], [S[i * n:(i + 1) * n] for i in range(ns)]
    d.hq, d.hs = [HQ[i * n:(i + 1) * n] for i in range(3)], [HS[i * n:(i + 1) * n] for i in range(ns)]
    d.txc = [X[i * m:(i + 1) * m] for i in range(9)]
    d._ptrs = None
    for t, u in zip(d.q + d.s, ref.q + ref.s):
        t.copy_(u)
    for t in d.hq + d.hs:
        t.fill_(3.0)            # the begin_step flag set before the search must still hold: the first substep overwrites this
    for k in range(3):          # (d.begin_step() is NOT called again)
        last = k == 2
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(1e-3 * d.kdt[k], 1.0 if last else d.kco[k], not last)
    ref.TIME_RUNGEKUTTA(1e-3)
    torch.cuda.synchronize()
    for t, u in zip(d.q + d.s + d.hq + d.hs, ref.q + ref.s + ref.hq + ref.hs):
        assert torch.equal(t, u)
    with pytest.raises(TlabError):          # the same allocation twice among the candidates
        bad = (c_vp * ncand)(*([blocks["hq"][0].data_ptr()] + [t.data_ptr() for t in blocks["q"][1:]]))
        check(load().tlab_dns_place_blocks(d._h, ncand, bad, arr["s"], arr["hq"], arr["hs"], arr["txc"], m, 1e-3, 1, 1, choice, rep), "tlab_dns_place_blocks")


def test_a_refused_allocation_leaves_no_error_behind(T):
    """TLab_AMD_Place_Arrays asks tlab_malloc for candidates until the memory says no: the refusal comes back as an error code and the NEXT launch's
    error check must not trip over it."""
    import ctypes
    import torch
    from tlab_amd.lib import load, c_vp
    L = load()
    p = c_vp(0)
    assert L.tlab_malloc(ctypes.byref(p), ctypes.c_size_t(1 << 42)) != 0          # 4 TiB
    n = 64
    g = T.FdmPlan(np.arange(n) / n, True, True)
    u = torch.rand(n * 8 * 8, dtype=torch.float64, device="cuda")
    r = torch.empty_like(u)
    T.OPR_Partial_X(T.OPR_P1, n, 8, 8, 0, g, u, r, None)                            # raises if the stale error surfaces
    torch.cuda.synchronize()
    assert bool(torch.isfinite(r).all())
