"""csrc/deferred.cpp through the C ABI: the recorded tail of a Runge-Kutta substep (tlab_deferred_*; include/tlab_amd.h) -- sequences that match
time.f90's become one fused substep, everything else runs literally in the order it came, and nothing is left behind when another entry point of
the library is called."""
import ctypes
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


def _dns(ns=1):
    from tlab_amd.dns import Dns
    nx, ny, nz = 256, 64, 32
    x = np.arange(nx) / nx
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    return Dns(x, y, z, nscal=ns, visc=1.0 / 500.0, schmidt=(0.7, 1.3)[:ns], yuniform=False, hyper_bc1_ext=0.1)


def _fields(d, seed):
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    Y = torch.linspace(0, 1, d.ny, dtype=torch.float64, device="cuda").view(1, d.ny, 1)
    wall = torch.sin(np.pi * Y)
    return [((2 * torch.rand(d.nz, d.ny, d.nx, dtype=torch.float64, device="cuda", generator=g) - 1) * wall).reshape(-1) for _ in range(3 + d.nscal)]


def _ptrs(d):
    from tlab_amd.lib import c_vp
    mk = lambda ts: (c_vp * max(1, len(ts)))(*[t.data_ptr() for t in ts])      # noqa: E731
    return mk(d.q), mk(d.s), mk(d.hq), mk(d.hs), mk(d.txc)


def _stats(L):
    c = (ctypes.c_longlong * 6)()
    assert L.tlab_deferred_stats(c) == 0
    return list(c)


KDT, KCO = [1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0], [-5.0 / 9.0, -153.0 / 128.0]


def _reference_step(d, f0, dt):
    import torch
    for t, a in zip(d.q + d.s, f0):
        t.copy_(a)
    d.begin_step()
    for k in range(3):
        d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dt * KDT[k], KCO[k] if k < 2 else 1.0, k < 2)
    torch.cuda.synchronize()
    return [t.clone() for t in d.q + d.s + d.hq + d.hs]


@pytest.mark.parametrize("ns", [1, 2])
def test_recorded_time_loop_equals_the_fused_substeps_bitwise(T, ns):
    """time.f90's own sequence -- zero fills, then per substep RHS, DAXPY x (3 + ns), DSCAL x (3 + ns) (none after the last) -- recorded, then a
    tlab_sync: three fused substeps ran, no BLAS pass, the zero fills were the begin_step; state AND tendencies equal the fused driver's."""
    import torch
    from tlab_amd.lib import load, check
    L = load()
    d = _dns(ns)
    f0 = _fields(d, 5)
    ref = _reference_step(d, f0, 2e-3)
    for t, a in zip(d.q + d.s, f0):
        t.copy_(a)
    for t in d.hq + d.hs:
        t.fill_(7.0)                       # garbage the zero fills must take care of
    torch.cuda.synchronize()
    q, s, hq, hs, txc = _ptrs(d)
    n = d.n
    before = _stats(L)
    check(L.tlab_deferred_enable(1), "enable")
    try:
        for t in d.hq + d.hs:
            check(L.tlab_deferred_zero(t.data_ptr(), n), "zero")
        for k in range(3):
            dte = 2e-3 * KDT[k]
            check(L.tlab_deferred_rhs(d._h, dte, q, s, hq, hs, txc), "rhs")
            for h, u in zip(d.hq + d.hs, d.q + d.s):
                check(L.tlab_deferred_axpy(n, dte, h.data_ptr(), u.data_ptr()), "axpy")
            if k < 2:
                for h in d.hq + d.hs:
                    check(L.tlab_deferred_scal(n, KCO[k], h.data_ptr()), "scal")
        check(L.tlab_sync(), "sync")
    finally:
        check(L.tlab_deferred_enable(0), "disable")
    after = _stats(L)
    assert [a - b for a, b in zip(after, before)] == [3, 0, 1, 0, 0, 0]
    for a, b in zip(d.q + d.s + d.hq + d.hs, ref):
        assert torch.equal(a, b)


def test_sequences_that_do_not_match_run_literally_in_order(T):
    """(a) another factor in one DAXPY, (b) a DSCAL before every field was updated, (c) an operator call in the middle of the tail: each time the
    fields equal those of the same calls executed one by one with the layer off."""
    import torch
    from tlab_amd.lib import load, check
    L = load()
    d = _dns(1)
    f0 = _fields(d, 9)
    q, s, hq, hs, txc = _ptrs(d)
    n, dte = d.n, 1e-3

    def run(variant, on):
        for t, a in zip(d.q + d.s, f0):
            t.copy_(a)
        for t in d.hq + d.hs:
            t.zero_()
        torch.cuda.synchronize()
        check(L.tlab_deferred_enable(1 if on else 0), "enable")
        try:
            check(L.tlab_deferred_rhs(d._h, dte, q, s, hq, hs, txc), "rhs")
            H, U = d.hq + d.hs, d.q + d.s
            if variant == "factor":
                for i, (h, u) in enumerate(zip(H, U)):
                    check(L.tlab_deferred_axpy(n, dte * (2.0 if i == 2 else 1.0), h.data_ptr(), u.data_ptr()), "axpy")
                for h in H:
                    check(L.tlab_deferred_scal(n, -0.5, h.data_ptr()), "scal")
            elif variant == "early scal":
                for h, u in list(zip(H, U))[:2]:
                    check(L.tlab_deferred_axpy(n, dte, h.data_ptr(), u.data_ptr()), "axpy")
                check(L.tlab_deferred_scal(n, -0.5, H[0].data_ptr()), "scal")
                for h, u in list(zip(H, U))[2:]:
                    check(L.tlab_deferred_axpy(n, dte, h.data_ptr(), u.data_ptr()), "axpy")
            else:
                for h, u in list(zip(H, U))[:3]:
                    check(L.tlab_deferred_axpy(n, dte, h.data_ptr(), u.data_ptr()), "axpy")
                T.OPR_Partial_X(T.OPR_P1, d.nx, d.ny, d.nz, 0, d.g[0], d.q[0], d.txc[5][: d.n], None)       # reads u: must see it updated
                check(L.tlab_deferred_axpy(n, dte, H[3].data_ptr(), U[3].data_ptr()), "axpy")
                for h in H:
                    check(L.tlab_deferred_scal(n, -0.5, h.data_ptr()), "scal")
            check(L.tlab_sync(), "sync")
        finally:
            check(L.tlab_deferred_enable(0), "disable")
        return [t.clone() for t in d.q + d.s + d.hq + d.hs + [d.txc[5][: d.n]]]

    for variant in ("factor", "early scal", "operator in between"):
        a, b = run(variant, True), run(variant, False)
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (variant, i)


def test_the_last_substep_of_a_step_waits_for_nothing(T):
    """After the last substep no DSCAL follows (time.f90:272): the description is completed by whatever comes next -- here a device-to-host copy of
    the field, which must already see the update."""
    import torch
    from tlab_amd.lib import load, check
    L = load()
    d = _dns(1)
    f0 = _fields(d, 13)
    for t, a in zip(d.q + d.s, f0):
        t.copy_(a)
    d.begin_step()
    d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(1e-3, 1.0, False)
    torch.cuda.synchronize()
    ref = d.q[1].cpu().numpy().copy()
    for t, a in zip(d.q + d.s, f0):
        t.copy_(a)
    torch.cuda.synchronize()
    q, s, hq, hs, txc = _ptrs(d)
    check(L.tlab_deferred_enable(1), "enable")
    try:
        for t in d.hq + d.hs:
            check(L.tlab_deferred_zero(t.data_ptr(), d.n), "zero")
        check(L.tlab_deferred_rhs(d._h, 1e-3, q, s, hq, hs, txc), "rhs")
        for h, u in zip(d.hq + d.hs, d.q + d.s):
            check(L.tlab_deferred_axpy(d.n, 1e-3, h.data_ptr(), u.data_ptr()), "axpy")
        out = np.empty(d.n)
        check(L.tlab_memcpy_d2h(out.ctypes.data, d.q[1].data_ptr(), d.n * 8), "d2h")
    finally:
        check(L.tlab_deferred_enable(0), "disable")
    assert np.array_equal(out, ref)
