"""ctypes wrapper around oracle/_ref/libtlab_ref.so (the reference's own Fortran, compiled in place).

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_golden.py and bench.py's cpu_baseline leg.
The library is global-state based (module variables in the reference), so one plan per direction at a time.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("TLAB_REF_LIB") or os.path.join(_HERE, "_ref", "libtlab_ref.so")      # TLAB_REF_LIB: another build of the same sources (_ref_omp, _ref_fma)

c_int, c_dbl = ctypes.c_int, ctypes.c_double
_P = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        # RTLD_LAZY: the IBM branch of opr_partial.f90 references ibm_spline_xyz_, never called here
        L = ctypes.CDLL(_PATH, mode=os.RTLD_LAZY)
        L.ref_init.argtypes = [c_int] * 3
        L.ref_fdm_create.argtypes = [c_int, c_int, _P, c_int, c_int, c_int, c_int]
        L.ref_fdm_info.argtypes = [c_int, c_int]
        L.ref_fdm_info.restype = c_int
        L.ref_fdm_get.argtypes = [c_int, c_int, _P, c_int]
        L.ref_der1_solve.argtypes = [c_int, c_int, c_int, _P, _P]
        L.ref_der2_solve.argtypes = [c_int, c_int, _P, _P, _P]
        L.ref_partial.argtypes = [c_int] * 6 + [_P, _P, _P]
        L.ref_transpose.argtypes = [_P, c_int, c_int, _P]
        L.ref_burgers.argtypes = [c_int] * 5 + [c_dbl, _P, _P, _P, _P]
        L.ref_filter_init.argtypes = [c_int] * 5 + [c_dbl, _P, c_dbl, _P, c_int, _P]
        L.ref_filter_1d.argtypes = [c_int] * 7 + [_P, _P, _P]
        L.ref_set_stagger.argtypes = [c_int]
        L.ref_intl_get.argtypes = [c_int, c_int, c_int, _P]
        _lib = L
    return _lib


def init(nx, ny, nz):
    lib().ref_init(nx, ny, nz)


def fdm_create(idir, nodes, periodic, uniform, mode1=6, mode2=7, hyper_bc1_ext=None):
    """FDM_CreatePlan of the reference.  hyper_bc1_ext: None = the plan as the reference builds it (the flang build reads 0.1 out of bounds for the
    extended wall-row entry of the default second derivative); a number = the second-derivative right-hand-side table and the Jacobians of the numpy
    oracle's plan at that closure written into the reference's plan (ref_driver.f90::ref_fdm_set_der2_rhs; every other table is checked to be the
    reference's own to the bit), so that the reference's solvers run on the closure the reference cannot build."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    lib().ref_fdm_create(idir, nodes.shape[0], nodes, int(periodic), int(uniform), mode1, mode2)
    if hyper_bc1_ext is not None and not periodic and mode2 == 7:
        import ctypes
        from . import tlab_oracle as O
        n = nodes.shape[0]
        o = O.FdmPlan(nodes, periodic, uniform, mode1, mode2, hyper_bc1_ext=hyper_bc1_ext)
        a = fdm_arrays(idir, n)
        for mine, theirs in ((o.der1.lhs, a["lhs1"]), (o.der1.rhs, a["rhs1"]), (o.der1.lu, a["lu1"]), (o.der2.lhs, a["lhs2"]), (o.der2.lu, a["lu2"]), (o.jac[:, 0], a["jac"][:, 0])):
            theirs = theirs[:, :mine.shape[1]] if mine.ndim == 2 else theirs
            # bitwise on the plain build; the build with fused multiply-adds rounds its own tables differently (<= 1e-14): there the written table is
            # the plain build's, i.e. the two builds of a closure-0.0 yardstick share this one table and differ in everything else
            if not np.array_equal(mine, theirs) and not np.abs(mine - theirs).max() <= 1e-13 * np.abs(theirs).max():
                raise RuntimeError("the numpy plan at closure %r differs from the reference's in a table the closure does not touch" % hyper_bc1_ext)
        rhs = np.asfortranarray(o.der2.rhs, dtype=np.float64)
        jac = np.asfortranarray(o.jac, dtype=np.float64)
        if rhs.shape != a["rhs2"].shape:
            raise RuntimeError("der2 rhs table: %r against the reference's %r" % (rhs.shape, a["rhs2"].shape))
        f = lib().ref_fdm_set_der2_rhs
        f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        f.restype = None
        f(idir, n, rhs.shape[1], rhs.ctypes.data, jac.ctypes.data)
        b = fdm_arrays(idir, n)
        if not (np.array_equal(b["rhs2"], o.der2.rhs) and np.array_equal(b["jac"], o.jac)):
            raise RuntimeError("ref_fdm_set_der2_rhs did not take")


def fdm_arrays(idir, n):
    """Plan arrays of direction idir as numpy arrays in the oracle's (row, diagonal) convention."""
    L = lib()
    info = {k: L.ref_fdm_info(idir, i) for i, k in enumerate(
        ["ndl1", "ndr1", "ndl2", "ndr2", "need_1der", "lu1_cols", "lu2_cols", "rhs2_cols"], start=1)}

    def get(which, rows, cols):
        buf = np.zeros(rows * cols)
        L.ref_fdm_get(idir, which, buf, buf.shape[0])
        return buf.reshape(cols, rows).T.copy()     # column-major -> [row, col]

    out = dict(info)
    out["lhs1"] = get(1, n, 5)
    out["rhs1"] = get(2, n, 7)
    out["lu1"] = get(3, n, info["lu1_cols"])
    out["rhs_b1"] = get(4, 4, 8)
    out["rhs_t1"] = get(5, 5, 7)
    out["mwn1"] = get(6, n, 1)[:, 0]
    out["lhs2"] = get(7, n, 5)
    out["rhs2"] = get(8, n, info["rhs2_cols"])
    out["lu2"] = get(9, n, info["lu2_cols"])
    out["mwn2"] = get(10, n, 1)[:, 0]
    out["jac"] = get(11, n, 3)
    return out


def der1_solve(idir, ibc, u):
    """u: (n, nlines) C-ordered == Fortran (nlines, n)."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    r = np.empty_like(u)
    lib().ref_der1_solve(idir, u.shape[1], ibc, u, r)
    return r


def der2_solve(idir, u, du):
    u = np.ascontiguousarray(u, dtype=np.float64)
    du = np.ascontiguousarray(du, dtype=np.float64)
    r = np.empty_like(u)
    lib().ref_der2_solve(idir, u.shape[1], u, du, r)
    return r


def partial(idir, itype, nx, ny, nz, ibc, u):
    u = np.ascontiguousarray(u, dtype=np.float64)
    r = np.zeros_like(u)
    t = np.zeros_like(u)
    lib().ref_partial(idir, itype, nx, ny, nz, ibc, u, r, t)
    return r, t


def transpose(a):
    """a: C-ordered (nca, nra) == Fortran a(nra, nca); returns C-ordered (nra, nca) == Fortran b(nca, nra)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    nca, nra = a.shape
    b = np.empty((nra, nca))
    lib().ref_transpose(a, nra, nca, b)
    return b


def burgers(idir, nx, ny, nz, ibc, visc, s, vel):
    s = np.ascontiguousarray(s, dtype=np.float64)
    vel = np.ascontiguousarray(vel, dtype=np.float64)
    r = np.zeros_like(s)
    t = np.zeros_like(s)
    lib().ref_burgers(idir, nx, ny, nz, ibc, float(visc), s, vel, r, t)
    return r, t


# ---- Poisson-path entry points (oracle/ref_driver_poisson.f90) -------------------------------------------
def _poisson_sigs():
    L = lib()
    if getattr(L, "_poisson_ready", False):
        return L
    L.ref_int1_create.argtypes = [c_dbl, c_int, c_int]
    L.ref_int1_get.argtypes = [c_int, c_int, _P, c_int]
    L.ref_int1_solve.argtypes = [c_int, c_int, _P, _P, _P]
    L.ref_ode2.argtypes = [c_int, c_int, c_dbl, _P, _P, _P, _P]
    L.ref_int2_create.argtypes = [c_dbl, c_int, c_int]
    L.ref_int2_get.argtypes = [c_int, _P, c_int]
    L.ref_int2_solve.argtypes = [c_int, _P, _P]
    L._poisson_ready = True
    return L


def int1_create(lam, ibc, factorize=True):
    """FDM_Int1_Initialize (factorize) or FDM_Int1_CreateSystem on the y plan (direction 2 must exist)."""
    _poisson_sigs().ref_int1_create(float(lam), int(ibc), int(factorize))


def int1_tables(n, ibc, nd_lhs=5, nd_rhs=3):
    """nd_lhs / nd_rhs: diagonals of the integral system / of its right-hand side = RHS / LHS diagonals of the derivative it inverts."""
    L = _poisson_sigs()

    def get(which, rows, cols):
        buf = np.zeros(rows * cols)
        L.ref_int1_get(int(ibc), which, buf, buf.shape[0])
        return buf.reshape(cols, rows).T.copy()

    return {"lhs": get(1, n, nd_lhs), "rhs": get(2, n, nd_rhs), "rhs_b": get(3, 5, 8), "rhs_t": get(4, 5, 8)}


def int1_solve(ibc, f, res):
    """f, res: (n, nlines) C-ordered; res carries the boundary value; returns (solution, du_boundary)."""
    L = _poisson_sigs()
    f = np.ascontiguousarray(f, dtype=np.float64)
    r = np.ascontiguousarray(res, dtype=np.float64).copy()
    du = np.zeros(f.shape[1])
    L.ref_int1_solve(int(ibc), f.shape[1], f, r, du)
    return r, du


def int2_create(lam2, ibc, factorize=True):
    """FDM_Int2_Initialize (factorize) or FDM_Int2_CreateSystem from the y plan's second derivative; ibc = BCS_DD/ND/DN/NN = 0..3."""
    _poisson_sigs().ref_int2_create(float(lam2), int(ibc), int(factorize))


def int2_tables(n, ndr=5, ndl=3):
    L = _poisson_sigs()

    def get(which, rows, cols):
        buf = np.zeros(rows * cols)
        L.ref_int2_get(which, buf, buf.shape[0])
        return buf.reshape(cols, rows).T.copy()

    return {"lhs": get(1, n, ndr), "rhs": get(2, n, ndl), "rhs_b": get(3, 5, 8), "rhs_t": get(4, 5, 8)}


def int2_solve(f, res):
    """f, res: (n, nlines) C-ordered; res carries the two boundary values; returns the solution."""
    L = _poisson_sigs()
    f = np.ascontiguousarray(f, dtype=np.float64)
    r = np.ascontiguousarray(res, dtype=np.float64).copy()
    L.ref_int2_solve(f.shape[1], f, r)
    return r


def ode2(itype, lam, f, bcs):
    """itype: 1 NN, 2 NN_Sing, 3 DD, 4 DD_Sing.  f: (n, nlines); bcs: (2, nlines).  Returns (u, v)."""
    L = _poisson_sigs()
    f = np.ascontiguousarray(f, dtype=np.float64).copy()
    b = np.ascontiguousarray(bcs, dtype=np.float64).copy()
    u = np.zeros_like(f)
    v = np.zeros_like(f)
    L.ref_ode2(int(itype), f.shape[1], float(lam), f, b, u, v)
    return u, v


def bcs_neumann_y(ibc, nx, ny, nz, u):
    """BOUNDARY_BCS_NEUMANN_Y on the y plan (direction 2): returns (bcs_hb, bcs_ht), each nx*nz."""
    L = lib()
    L.ref_bcs_neumann_y.argtypes = [c_int] * 4 + [_P, _P, _P]
    u = np.ascontiguousarray(u, dtype=np.float64)
    hb = np.zeros(nx * nz)
    ht = np.zeros(nx * nz)
    L.ref_bcs_neumann_y(int(ibc), nx, ny, nz, u, hb, ht)
    return hb, ht


def grid_write(name, x, y, z):
    L = lib()
    x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
    L.ref_grid_write.argtypes = [ctypes.c_char_p, c_int, c_int, c_int, _P, _P, _P]
    L.ref_grid_write(name.encode(), x.size, y.size, z.size, x, y, z)


def grid_read(name, nx, ny, nz):
    L = lib()
    x, y, z, sc = np.zeros(nx), np.zeros(ny), np.zeros(nz), np.zeros(3)
    L.ref_grid_read.argtypes = [ctypes.c_char_p, c_int, c_int, c_int, _P, _P, _P, _P]
    L.ref_grid_read(name.encode(), nx, ny, nz, x, y, z, sc)
    return x, y, z, sc


def io_write_fields(name, nx, ny, nz, nt, fields, params=()):
    L = lib()
    a = np.ascontiguousarray(np.stack([np.asarray(f, dtype=np.float64).reshape(-1) for f in fields]))      # (nfield, n) = Fortran a(n, nfield)
    p = np.ascontiguousarray(params if len(params) else [0.0], dtype=np.float64)
    L.ref_io_write_fields.argtypes = [ctypes.c_char_p] + [c_int] * 5 + [_P, c_int, _P]
    L.ref_io_write_fields(name.encode(), nx, ny, nz, nt, len(fields), a, len(params), p)


def io_read_fields(name, nx, ny, nz, nt, nfield, nparams):
    L = lib()
    a = np.zeros((nfield, nx * ny * nz))
    p = np.zeros(max(nparams, 1))
    L.ref_io_read_fields.argtypes = [ctypes.c_char_p] + [c_int] * 5 + [_P, c_int, _P]
    L.ref_io_read_fields(name.encode(), nx, ny, nz, nt, nfield, a, nparams, p)
    return [a[i] for i in range(nfield)], p[:nparams]


# 1-D filters (src/filters/*.f90 through oracle/ref_driver_filter.f90): DNS_FILTER_COMPACT = 1, _6E = 2, _4E = 3, _COMPACT_CUTOFF = 9
FILTER_NCOLS = {1: 10, 2: 0, 3: 5, 9: 7}        # inb_filter (opr_filter.f90:121-139)


def filter_init(itype, nodes, jac, periodic, bcsmin=1, bcsmax=1, alpha=0.49, scale=None):
    """OPR_FILTER_INITIALIZE (opr_filter.f90:236-275) -> coeffs as [row, column] (n, inb_filter)."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    jac = np.ascontiguousarray(jac, dtype=np.float64)
    n, nc = nodes.shape[0], max(FILTER_NCOLS[itype], 1)
    if scale is None:
        scale = (nodes[-1] - nodes[0]) * (n / (n - 1.0) if periodic else 1.0)
    buf = np.zeros(n * nc)
    lib().ref_filter_init(itype, n, int(periodic), bcsmin, bcsmax, float(alpha), jac, float(scale), nodes, nc, buf)
    return buf.reshape(nc, n).T.copy()[:, :FILTER_NCOLS[itype]]


def filter_1d(itype, periodic, bcsmin, bcsmax, coeffs, u):
    """OPR_FILTER_1D (opr_filter.f90:393-460).  u: (n, nlines) C-ordered == Fortran (nlines, n)."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    n = u.shape[0]
    nc = max(FILTER_NCOLS[itype], 1)
    c = np.zeros((nc, n))
    if FILTER_NCOLS[itype]:
        c[:, :] = np.asarray(coeffs, dtype=np.float64).T
    r = np.empty_like(u)
    lib().ref_filter_1d(itype, n, u.shape[1], int(periodic), bcsmin, bcsmax, nc, np.ascontiguousarray(c.reshape(-1)), u, r)
    return r


def set_stagger(on):
    """TLab_WorkFlow::stagger_on for the plans created afterwards (fdm.f90:236-248)."""
    lib().ref_set_stagger(int(bool(on)))


def intl_arrays(idir, n):
    """g%intl%lu0i, g%intl%lu1i as [row, column] (n, 5)"""
    out = []
    for which in (0, 1):
        buf = np.zeros(n * 5)
        lib().ref_intl_get(idir, which, n, buf)
        out.append(buf.reshape(5, n).T.copy())
    return out
