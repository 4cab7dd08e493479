"""CPU tests: the numpy oracle (oracle/tlab_oracle.py) is pinned
 (1) against the golden vectors generated from the reference's own Fortran (tests/golden/make_golden.py), and
 (2) against oracle/_ref/libtlab_ref.so directly when it is present (build container, or travelled to the GPU box).
Agreement is expected to ~1e-15 (observed: bitwise); the asserted bound is 1e-14."""
import numpy as np
import pytest
from conftest import golden_files, rel_err
from oracle import tlab_oracle as O
from oracle import ref_lib as R

TOL = 1e-14
PLAN_KEYS = [("lhs1", "der1", "lhs"), ("rhs1", "der1", "rhs"), ("lu1", "der1", "lu"), ("rhs_b1", "der1", "rhs_b"),
             ("rhs_t1", "der1", "rhs_t"), ("mwn1", "der1", "mwn"), ("lhs2", "der2", "lhs"), ("rhs2", "der2", "rhs"),
             ("lu2", "der2", "lu"), ("mwn2", "der2", "mwn")]


def plans_from_golden(g):
    spec = {1: (g["x"], True, True), 2: (g["y"], False, bool(g["yuniform"])), 3: (g["z"], True, True)}
    return {d: O.FdmPlan(n, p, u, int(g["mode1"]), int(g["mode2"])) for d, (n, p, u) in spec.items()}


@pytest.mark.parametrize("path", golden_files("derivs_"))
def test_plan_arrays_match_golden(path):
    g = np.load(path)
    plans = plans_from_golden(g)
    for d in (1, 2, 3):
        p = plans[d]
        for key, der, attr in PLAN_KEYS:
            assert rel_err(getattr(getattr(p, der), attr), g["plan%d_%s" % (d, key)]) <= TOL, (d, key)
        assert rel_err(p.jac, g["plan%d_jac" % d]) <= TOL
        assert int(p.der2.need_1der) == int(g["plan%d_need_1der" % d])
        assert p.der1.nb_diag == (int(g["plan%d_ndl1" % d]), int(g["plan%d_ndr1" % d]))
        assert p.der2.nb_diag == (int(g["plan%d_ndl2" % d]), int(g["plan%d_ndr2" % d]))


@pytest.mark.parametrize("path", golden_files("derivs_"))
def test_operators_match_golden(path):
    g = np.load(path)
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    plans = plans_from_golden(g)
    u, v, visc = g["u"], g["v"], float(g["visc"])
    for d in (1, 2, 3):
        for ibc in ((0, 1, 2, 3) if d == 2 else (0,)):
            for t in (O.OPR_P1, O.OPR_P2, O.OPR_P2_P1):
                r, t1 = O.opr_partial(d, t, nx, ny, nz, ibc, plans[d], u)
                assert rel_err(r, g["partial_d%d_t%d_bc%d" % (d, t, ibc)]) <= TOL
                if t == O.OPR_P2_P1:
                    assert rel_err(t1, g["partial_d%d_t%d_bc%d_tmp1" % (d, t, ibc)]) <= TOL
            r, st = O.opr_burgers(d, nx, ny, nz, ibc, plans[d], visc, u, v)
            assert rel_err(r, g["burgers_d%d_bc%d" % (d, ibc)]) <= TOL
            if d != 3:
                # transposed operand the reference leaves in tmp1: pure index work, bit-exact
                assert np.array_equal(st, g["burgers_d%d_bc%d_tmp1" % (d, ibc)])


@pytest.mark.parametrize("path", golden_files("derivs_"))
def test_boundary_bcs_neumann_matches_golden(path):
    g = np.load(path)
    if "bcsn_bc1_hb" not in g.files:
        pytest.skip("no BOUNDARY_BCS_NEUMANN_Y vectors in this file (the routine is written for the CompactJacobian6 first derivative)")
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    gy = plans_from_golden(g)[2]
    for ibc in (1, 2, 3):
        hb, ht = O.boundary_bcs_neumann_y(ibc, nx, ny, nz, gy, g["u"])
        assert rel_err(hb.ravel(), g["bcsn_bc%d_hb" % ibc]) <= TOL if ibc & 1 else not hb.any()
        assert rel_err(ht.ravel(), g["bcsn_bc%d_ht" % ibc]) <= TOL if ibc & 2 else not ht.any()
    # the property the routine exists for: with these wall values the y-derivative vanishes at the walls
    a = g["u"].reshape(nz, ny, nx).copy()
    a[:, 0, :] = g["bcsn_bc3_hb"].reshape(nz, nx)
    a[:, -1, :] = g["bcsn_bc3_ht"].reshape(nz, nx)
    d = O.opr_partial(2, O.OPR_P1, nx, ny, nz, 0, gy, a.ravel())[0].reshape(nz, ny, nx)
    assert np.abs(d[:, 0, :]).max() <= 1e-11 * np.abs(d).max() and np.abs(d[:, -1, :]).max() <= 1e-11 * np.abs(d).max()


def test_hyper_wall_closure_defect_is_what_the_reference_does():
    """DESIGN.md 'reference defects': the flang-built reference reads coef_bc1(7) out of bounds and gets 0.1."""
    g = np.load(golden_files("derivs_stretched")[0])
    rhs2 = g["plan2_rhs2"]
    coef3 = rhs2[0, 3] / 13.0            # rows are normalised by 1/coef_int(3): entry 13/coef(3)
    assert abs(rhs2[0, 0] / coef3 - O.HYPER_BC1_EXT) < 1e-14


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("n,periodic,stretch", [(16, True, False), (33, False, True), (40, False, False), (64, True, False)])
def test_oracle_vs_reference_library_1d(n, periodic, stretch):
    rng = np.random.default_rng(n)
    if periodic:
        nodes = np.arange(n) / n * 3.0
    elif stretch:
        nodes = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(1.5))
    else:
        nodes = np.arange(n) / (n - 1)
    R.init(n, n, n)
    R.fdm_create(2, nodes, periodic, not stretch)
    p = O.FdmPlan(nodes, periodic, not stretch)
    u = rng.uniform(-1, 1, (n, 7))
    for ibc in ((0,) if periodic else (0, 1, 2, 3)):
        d1 = O.der1_solve(p.der1, ibc, u)
        assert rel_err(d1, R.der1_solve(2, ibc, u)) <= TOL
        assert rel_err(O.der2_solve(p.der2, p.der2.lu, u, d1), R.der2_solve(2, u, d1)) <= TOL


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_transpose_bit_exact_vs_reference():
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (70, 130))           # Fortran a(130, 70): spans the 64x64 blocking of TLab_Transpose
    assert np.array_equal(R.transpose(a), np.ascontiguousarray(a.T))
