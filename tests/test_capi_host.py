"""CPU tests of the product's host side (no GPU, no compute kernels):
 - the C-ABI library loads and exports every symbol include/tlab_amd.h declares,
 - its own plan generator (C++ restatement of FDM_CreatePlan) reproduces the reference's tables (golden vectors),
 - the chunked-Thomas tables the kernels consume solve the systems (host emulation of the device algorithm),
 - operators fail loudly without a GPU (no CPU fallback)."""
import os
import re
import numpy as np
import pytest
from conftest import golden_files, rel_err, ROOT
import tlab_amd as T
from tlab_amd import lib as L
from oracle import tlab_oracle as O

KEYS = ["lhs1", "rhs1", "lu1", "rhs_b1", "rhs_t1", "mwn1", "lhs2", "rhs2", "lu2", "mwn2", "jac"]


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "tlab_amd.h")).read()
    declared = set(re.findall(r"\b(tlab_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("tlab_fdm_plan")
    lib = L.load()
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("path", golden_files("derivs_"))
def test_plan_generator_matches_reference_tables(path):
    g = np.load(path)
    spec = {1: (g["x"], True, True), 2: (g["y"], False, bool(g["yuniform"])), 3: (g["z"], True, True)}
    for d, (nodes, per, uni) in spec.items():
        p = T.FdmPlan(nodes, per, uni, int(g["mode1"]), int(g["mode2"]))
        for k in KEYS:
            assert rel_err(p.table(k), g["plan%d_%s" % (d, k)]) <= 1e-14, (d, k)
        assert p.need_1der == bool(g["plan%d_need_1der" % d])
        assert (p.info(1), p.info(2), p.info(3), p.info(4)) == tuple(int(g["plan%d_%s" % (d, k)]) for k in ("ndl1", "ndr1", "ndl2", "ndr2"))


def test_hyper_closure_switch():
    y = np.arange(40) / 39.0
    a = T.FdmPlan(y, False, True, hyper_bc1_ext=0.1).table("rhs2")
    b = T.FdmPlan(y, False, True, hyper_bc1_ext=0.0).table("rhs2")
    assert b[0, 0] == 0.0 and b[-1, 6] == 0.0 and a[0, 0] != 0.0
    assert np.array_equal(a[1:-1, :7], b[1:-1, :7])    # (the defect also leaks into d2x/ds2, i.e. the correction columns)


@pytest.mark.parametrize("n,periodic,stretch,chunks", [
    (64, True, False, 1), (128, True, False, 64), (256, True, False, 64), (256, True, False, 8),
    (96, False, True, 3), (128, False, True, 64), (128, False, False, 4), (512, False, True, 16), (50, False, True, 1),
    (1024, True, False, 128), (2048, True, False, 256)])      # two / four waves per x line (k_xline WPL = 2, 4)
def test_device_tables_solve_the_systems(n, periodic, stretch, chunks):
    """Host emulation of the kernels' chunked Thomas (same tables, same operation order) vs the oracle's TRIDSS/TRIDPSS."""
    rng = np.random.default_rng(n + chunks)
    if periodic:
        nodes = np.arange(n) / n
    elif stretch:
        nodes = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2))
    else:
        nodes = np.arange(n) / (n - 1)
    p = T.FdmPlan(nodes, periodic, not stretch)
    o = O.FdmPlan(nodes, periodic, not stretch)
    f = rng.uniform(-1, 1, n)
    for ibc in ((0,) if periodic else (0, 1, 2, 3)):
        ref = f.copy().reshape(n, 1)
        nmin, nmax = 0, n
        if not periodic:
            if ibc in (1, 3):
                ref[0] = 0.0; nmin = 1
            if ibc in (2, 3):
                ref[n - 1] = 0.0; nmax = n - 1
            ip = ibc * 5
            O.tridss(o.der1.lu[nmin:nmax, ip], o.der1.lu[nmin:nmax, ip + 1], o.der1.lu[nmin:nmax, ip + 2], ref[nmin:nmax])
        else:
            O.tridpss(*(o.der1.lu[:, k] for k in range(5)), ref)
        fin = f.copy()
        if not periodic:
            if ibc in (1, 3): fin[0] = 0.0
            if ibc in (2, 3): fin[n - 1] = 0.0
        got = p.debug_host_chunked_solve(1, ibc, chunks, fin)
        assert rel_err(got, ref[:, 0]) <= 1e-13, ibc
    ref = f.copy().reshape(n, 1)
    if periodic:
        O.tridpss(*(o.der2.lu[:, k] for k in range(5)), ref)
    else:
        O.tridss(o.der2.lu[:, 0], o.der2.lu[:, 1], o.der2.lu[:, 2], ref)
    assert rel_err(p.debug_host_chunked_solve(2, 0, chunks, f), ref[:, 0]) <= 1e-13


def test_from_arrays_roundtrip():
    g = np.load(golden_files("derivs_stretched")[0])
    n = int(g["ny"])
    p = T.FdmPlan.from_arrays(n, False, int(g["plan2_need_1der"]), g["plan2_lhs1"], g["plan2_rhs1"][:, :5],
                              g["plan2_lhs2"], g["plan2_rhs2"][:, :10])
    assert rel_err(p.table("rhs2"), g["plan2_rhs2"]) == 0.0 and rel_err(p.table("lhs1"), g["plan2_lhs1"]) == 0.0
    assert p.need_1der


def test_penta_plan_tables_match_the_reference():
    """SpaceOrder1 = CompactJacobian6Penta: the C++ restatement (fdm_schemes.cpp: FDM_C1N6_Jacobian_Penta, PENTADFS2, PENTADPFS, the
    out-of-bounds wall coefficient included) against the tables the reference itself produced (tests/golden/derivs_penta_*.npz)."""
    g = np.load(golden_files("derivs_penta")[0])
    for d, (nodes, per, uni) in {1: (g["x"], True, True), 2: (g["y"], False, False), 3: (g["z"], True, True)}.items():
        p = T.FdmPlan(nodes, per, uni, scheme1=5, scheme2=6)
        for key in ("lhs1", "rhs1", "lu1", "rhs_b1", "rhs_t1", "mwn1", "lhs2", "rhs2", "lu2", "jac"):
            a, b = np.asarray(p.table(key)), g["plan%d_%s" % (d, key)]
            if a.ndim == 2:
                m = min(a.shape[1], b.shape[1])
                a, b = a[:, :m], b[:, :m]
            assert rel_err(a, b) <= 1e-14, (d, key)
    # the host's own tables through from_arrays: the library factorizes them the reference's way
    n = int(g["ny"])
    q = T.FdmPlan.from_arrays(n, False, int(g["plan2_need_1der"]), g["plan2_lhs1"], g["plan2_rhs1"][:, :7], g["plan2_lhs2"], g["plan2_rhs2"][:, :10], ndl1=5)
    assert rel_err(q.table("lu1"), g["plan2_lu1"]) <= 1e-14 and rel_err(q.table("rhs_t1"), g["plan2_rhs_t1"]) <= 1e-14


def test_unsupported_and_invalid_are_reported():
    with pytest.raises(T.TlabError):
        T.FdmPlan(np.arange(32) / 31.0, False, True, scheme1=16)        # direct FIRST derivatives (fdm_comx_direct.f90): not built
    with pytest.raises(T.TlabError):
        T.FdmPlan(np.arange(32) ** 1.5, True, False)                    # periodic must be uniform (fdm.f90:117)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(T.TlabError):
        T.init(0)
    p = T.FdmPlan(np.arange(64) / 64.0, True, True)
    rc = L.load().tlab_opr_partial(1, p._h, 1, 64, 1, 1, 0, L.c_vp(8), L.c_vp(16), L.c_vp(0))
    assert rc != 0       # refused: tlab_init never succeeded


def test_host_factorized_int1_tables_equal_the_oracle_bitwise():
    """FDM_Int1 with 3 and 7 diagonals (SpaceOrder1 = CompactJacobian4 / CompactJacobian6Penta under the factorized Poisson solver): the library builds
    and factorizes these systems on the HOST (tlab_amd/csrc/int1_generic.cpp: FDM_Int1_CreateSystem, FDM_Bcs_Reduce, TRIDFS / HEPTADFS); its tables
    equal the oracle's to the bit, and the oracle's equal the reference's (tests/test_oracle_poisson.py on the _ref-generated fixtures)."""
    import ctypes
    import numpy as np
    import tlab_amd as T
    from tlab_amd.lib import load, check
    from oracle import tlab_oracle as O, tlab_oracle_poisson as OP
    L = load()
    dp = ctypes.POINTER(ctypes.c_double)
    for n, stretch in ((40, True), (33, False)):
        y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2)) if stretch else np.arange(n) / (n - 1.0) * 2.0
        for mode1 in (5, 4):
            gp, op = T.FdmPlan(y, False, not stretch, mode1, 7), O.FdmPlan(y, False, not stretch, mode1, 7)
            ndl, ndr = op.der1.nb_diag
            lam = np.array([0.0, 1.2246467991473532e-16, 0.5, 6.283185307179586, 97.0, 1500.0])
            nm = len(lam)
            for ibc, sgn in ((1, 1.0), (2, -1.0)):
                fac, rb, rt, R = np.zeros((ndr, n, nm)), np.zeros((40, nm)), np.zeros((40, nm)), np.zeros((n, ndl))
                ls = np.ascontiguousarray(sgn * lam)
                check(L.tlab_debug_int1_tables(gp._h, ibc, nm, ls.ctypes.data_as(dp), fac.ctypes.data_as(dp), rb.ctypes.data_as(dp),
                                               rt.ctypes.data_as(dp), R.ctypes.data_as(dp)), "tlab_debug_int1_tables")
                p = OP.int1_initialize(op.der1, ls, ibc)
                assert np.array_equal(fac.transpose(1, 0, 2), p.lhs), (n, mode1, ibc, "factors")
                assert np.array_equal(rb, p.rhs_b.transpose(1, 0, 2).reshape(40, nm)) and np.array_equal(rt, p.rhs_t.transpose(1, 0, 2).reshape(40, nm))
                assert np.array_equal(R, p.rhs)
