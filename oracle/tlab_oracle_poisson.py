"""CPU oracle, part 2: numpy restatement of OPR_Poisson_FourierXZ_Factorize and what it stands on.

TEST INFRASTRUCTURE ONLY (same rules as tlab_oracle.py).  Parity status: PINNED for the per-mode arithmetic
(FDM_Int1_*, PENTADFS/SS, OPR_ODE2_Factorize_*) against oracle/_ref (the reference's own Fortran) and the golden
vectors generated from it; the FFTs are numpy.fft (the reference uses whichever FFTW3 provider it is linked with,
un-vendored; FFT parity is pinned to ~1e-15 relative, not bitwise -- SURVEY.md 8c).

Vectorisation: every per-mode quantity carries a trailing mode axis M, so the whole (kx,kz) plane is solved at once.
  lambda: (M,)   lhs: (n, 5, M)   rhs: (n, 3)   rhs_b: (5, 8, M) [row-1, col]   rhs_t: (5, 8, M) [row, col-1]
  lines:  (n, nlines, M)          boundary values: (nlines, M)
"""
import numpy as np

from .tlab_oracle import BCS_MIN, BCS_MAX, BCS_BOTH, BCS_NN, BCS_DD, BCS_ND, BCS_DN


# ######################################################################################
# utils/linear5.f90
# ######################################################################################
def pentadfs(a, b, c, d, e):
    """utils/linear5.f90:30-71 PENTADFS, in place; arrays (nmax, M)."""
    nmax = a.shape[0]
    b[1] = b[1] / c[0]
    c[1] = c[1] - b[1] * d[0]
    d[1] = d[1] - b[1] * e[0]
    for n in range(2, nmax - 1):
        a[n] = a[n] / c[n - 2]
        b[n] = (b[n] - a[n] * d[n - 2]) / c[n - 1]
        c[n] = c[n] - b[n] * d[n - 1] - a[n] * e[n - 2]
        d[n] = d[n] - b[n] * e[n - 1]
    n = nmax - 1
    a[n] = a[n] / c[n - 2]
    b[n] = (b[n] - a[n] * d[n - 2]) / c[n - 1]
    c[n] = c[n] - b[n] * d[n - 1] - a[n] * e[n - 2]
    a[2:] = -a[2:]
    b[1:] = -b[1:]
    c[:] = 1.0 / c
    d[:nmax - 1] = -d[:nmax - 1]
    e[:nmax - 2] = -e[:nmax - 2]


def pentadss(a, b, c, d, e, f):
    """utils/linear5.f90:76-131 PENTADSS; coefficients (nmax, M), f (nmax, nlines, M) in place."""
    nmax = a.shape[0]
    f[1] = f[1] + f[0] * b[1]
    for n in range(2, nmax):
        f[n] = f[n] + f[n - 1] * b[n] + f[n - 2] * a[n]
    n = nmax - 1
    f[n] = f[n] * c[n]
    n = nmax - 2
    f[n] = (f[n] + f[n + 1] * d[n]) * c[n]
    for n in range(nmax - 3, -1, -1):
        f[n] = (f[n] + f[n + 1] * d[n] + f[n + 2] * e[n]) * c[n]


# ######################################################################################
# utils/linear3.f90, utils/linear7.f90: the band solvers of the 3- and 7-diagonal integral systems (fdm_integral.f90:75-83, 255-263)
# ######################################################################################
def tridfs_m(a, b, c):
    """utils/linear3.f90:29-51 TRIDFS, in place; arrays (nmax, M)."""
    nmax = a.shape[0]
    for n in range(1, nmax):
        a[n] = a[n] / b[n - 1]
        b[n] = b[n] - a[n] * c[n - 1]
    a[:] = -a
    b[:] = 1.0 / b
    c[:] = -c


def tridss_m(a, b, c, f):
    """utils/linear3.f90:56-150 TRIDSS; coefficients (nmax, M), f (nmax, nlines, M) in place."""
    nmax = a.shape[0]
    for n in range(1, nmax):
        f[n] = f[n] + a[n] * f[n - 1]
    f[nmax - 1] = f[nmax - 1] * b[nmax - 1]
    for n in range(nmax - 2, -1, -1):
        f[n] = (f[n] + c[n] * f[n + 1]) * b[n]


def heptadfs(a, b, c, d, e, f, g):
    """utils/linear7.f90:30-93 HEPTADFS, in place; arrays (nmax, M) (0-based rows)."""
    nmax = a.shape[0]
    g[0] = g[0] / d[0]
    f[0] = f[0] / d[0]
    e[0] = e[0] / d[0]
    c[0] = 1.0 / d[0]
    d[0] = 1.0
    c[1] = c[1] / d[0]
    d[1] = d[1] - c[1] * e[0]
    e[1] = e[1] - c[1] * f[0]
    f[1] = f[1] - c[1] * g[0]
    b[2] = b[2] / d[0]
    c[2] = (c[2] - b[2] * e[0]) / d[1]
    d[2] = d[2] - c[2] * e[1] - b[2] * f[0]
    e[2] = e[2] - c[2] * f[1] - b[2] * g[0]
    f[2] = f[2] - c[2] * g[1]
    for n in range(3, nmax - 2):
        a[n] = a[n] / d[n - 3]
        b[n] = (b[n] - a[n] * e[n - 3]) / d[n - 2]
        c[n] = (c[n] - b[n] * e[n - 2] - a[n] * f[n - 3]) / d[n - 1]
        d[n] = d[n] - c[n] * e[n - 1] - b[n] * f[n - 2] - a[n] * g[n - 3]
        e[n] = e[n] - c[n] * f[n - 1] - b[n] * g[n - 2]
        f[n] = f[n] - c[n] * g[n - 1]
    n = nmax - 2
    a[n] = a[n] / d[n - 3]
    b[n] = (b[n] - a[n] * e[n - 3]) / d[n - 2]
    c[n] = (c[n] - b[n] * e[n - 2] - a[n] * f[n - 3]) / d[n - 1]
    d[n] = d[n] - c[n] * e[n - 1] - b[n] * f[n - 2] - a[n] * g[n - 3]
    e[n] = e[n] - c[n] * f[n - 1] - b[n] * g[n - 2]
    n = nmax - 1
    a[n] = a[n] / d[n - 3]
    b[n] = (b[n] - a[n] * e[n - 3]) / d[n - 2]
    c[n] = (c[n] - b[n] * e[n - 2] - a[n] * f[n - 3]) / d[n - 1]
    d[n] = d[n] - c[n] * e[n - 1] - b[n] * f[n - 2] - a[n] * g[n - 3]


def heptadss(a, b, c, d, e, f, g, frc):
    """utils/linear7.f90:98-142 HEPTADSS; coefficients (nmax, M), frc (nmax, nlines, M) in place."""
    nmax = a.shape[0]
    frc[0] = frc[0] * c[0]
    frc[1] = frc[1] - frc[0] * c[1]
    frc[2] = frc[2] - frc[1] * c[2] - frc[0] * b[2]
    for n in range(3, nmax):
        frc[n] = frc[n] - frc[n - 1] * c[n] - frc[n - 2] * b[n] - frc[n - 3] * a[n]
    n = nmax - 1
    frc[n] = frc[n] / d[n]
    frc[n - 1] = (frc[n - 1] - frc[n] * e[n - 1]) / d[n - 1]
    frc[n - 2] = (frc[n - 2] - frc[n - 1] * e[n - 2] - frc[n] * f[n - 2]) / d[n - 2]
    for n in range(nmax - 4, -1, -1):
        frc[n] = (frc[n] - frc[n + 1] * e[n] - frc[n + 2] * f[n] - frc[n + 3] * g[n]) / d[n]


# ######################################################################################
# fdm/fdm_base.f90:304-391 FDM_Bcs_Reduce
# lhs (nx, ndl[, M]) modified in place; rhs (nx, ndr[, M]); rhs_b (>=4, 8[, M]) [row-1, col]; rhs_t (5, 8[, M]) [row, col-1]
# ######################################################################################
def fdm_bcs_reduce(ibc, lhs, rhs=None, rhs_b=None, rhs_t=None):
    ndl = lhs.shape[1]
    idl = ndl // 2 + 1
    nx = lhs.shape[0]
    if rhs is not None:
        ndr = rhs.shape[1]
        idr = ndr // 2 + 1
        nx_t = idr
        mx = max(idl, idr + 1)
    if ibc in (BCS_MIN, BCS_BOTH):
        dummy = 1.0 / lhs[0, idl - 1]
        lhs[0, :] = -lhs[0, :] * dummy
        lhs[0, idl - 1] = 1.0
        for ir in range(1, idl):
            for ic in range(idl + 1, ndl + 1):
                lhs[ir, ic - ir - 1] = lhs[ir, ic - ir - 1] + lhs[ir, idl - ir - 1] * lhs[0, ic - 1]
            ic = ndl + 1
            lhs[ir, ic - ir - 1] = lhs[ir, ic - ir - 1] + lhs[ir, idl - ir - 1] * lhs[0, 0]
        if rhs_b is not None:
            rhs_b[0:mx, 1:ndr + 1] = rhs[0:mx, 0:ndr]
            rhs_b[0, 1:ndr + 1] = rhs_b[0, 1:ndr + 1] * dummy
            for ir in range(1, idl):
                for ic in range(idr, ndr + 1):
                    rhs_b[ir, ic - ir] = rhs_b[ir, ic - ir] - lhs[ir, idl - ir - 1] * rhs_b[0, ic]
                ic = ndr + 1
                rhs_b[ir, ic - ir] = rhs_b[ir, ic - ir] - lhs[ir, idl - ir - 1] * rhs_b[0, 1]
    if ibc in (BCS_MAX, BCS_BOTH):
        dummy = 1.0 / lhs[nx - 1, idl - 1]
        lhs[nx - 1, :] = -lhs[nx - 1, :] * dummy
        lhs[nx - 1, idl - 1] = 1.0
        for ir in range(1, idl):
            # ic = 0: longer stencil at the boundary
            lhs[nx - ir - 1, ir - 1] = lhs[nx - ir - 1, ir - 1] + lhs[nx - ir - 1, idl + ir - 1] * lhs[nx - 1, ndl - 1]
            for ic in range(1, idl):
                lhs[nx - ir - 1, ic + ir - 1] = lhs[nx - ir - 1, ic + ir - 1] + lhs[nx - ir - 1, idl + ir - 1] * lhs[nx - 1, ic - 1]
        if rhs_t is not None:
            rhs_t[nx_t - mx + 1:nx_t + 1, 0:ndr] = rhs[nx - mx:nx, 0:ndr]
            rhs_t[nx_t, 0:ndr] = rhs_t[nx_t, 0:ndr] * dummy
            for ir in range(1, idl):
                rhs_t[nx_t - ir, ir - 1] = rhs_t[nx_t - ir, ir - 1] - lhs[nx - ir - 1, idl + ir - 1] * rhs_t[nx_t, ndr - 1]
                for ic in range(1, idr + 1):
                    rhs_t[nx_t - ir, ic + ir - 1] = rhs_t[nx_t - ir, ic + ir - 1] - lhs[nx - ir - 1, idl + ir - 1] * rhs_t[nx_t, ic - 1]


# ######################################################################################
# fdm/fdm_integral.f90
# ######################################################################################
class Int1Plan:
    """fdm/fdm_integral.f90:18-26 type fdm_integral_dt, for a vector of lambdas (trailing axis M)."""
    pass


def int1_create_system(g, lam, ibc):
    """fdm/fdm_integral.f90:91-214 FDM_Int1_CreateSystem.  g: oracle DerPlan of the first derivative; lam: (M,)."""
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64))
    M = lam.shape[0]
    ndl, ndr = g.nb_diag
    idl, idr = ndl // 2 + 1, ndr // 2 + 1
    nx = g.size
    p = Int1Plan()
    p.lam, p.bc, p.nx, p.ndl, p.ndr = lam, ibc, nx, ndr, ndl        # NB: lhs of the integral has ndr diagonals, rhs ndl
    A = g.lhs[:, :ndl].copy()                                        # fdmi%rhs
    rhsr_b = np.zeros((5, 8))
    rhsr_t = np.zeros((5, 8))
    fdm_bcs_reduce(ibc, A, g.rhs[:, :ndr], rhsr_b, rhsr_t)
    rhs_b = np.zeros((5, 8))
    rhs_t = np.zeros((5, 8))
    if ibc == BCS_MIN:
        rhs_b[0:idl + 1, 1:ndl + 1] = A[0:idl + 1, 0:ndl]
        for ir in range(1, idr):
            rhs_b[ir, idl - ir] = -rhsr_b[ir, idr - ir]
    else:
        rhs_t[0:idl + 1, 0:ndl] = A[nx - idl - 1:nx, 0:ndl]
        for ir in range(1, idr):
            rhs_t[idl - ir, idl + ir - 1] = -rhsr_t[idr - ir, idr + ir - 1]
    # new lhs diagonals C = B + lambda A (lambda-dependent), :150-156
    lhs = np.repeat(g.rhs[:, :ndr, None], M, axis=2).astype(np.float64)
    lhs[:, idr - 1, :] = lhs[:, idr - 1, :] + lam * g.lhs[:, idl - 1, None]
    for i in range(1, idl):
        lhs[i:nx, idr - i - 1, :] = lhs[i:nx, idr - i - 1, :] + lam * g.lhs[i:nx, idl - i - 1, None]
        lhs[0:nx - i, idr + i - 1, :] = lhs[0:nx - i, idr + i - 1, :] + lam * g.lhs[0:nx - i, idl + i - 1, None]
    if ibc == BCS_MIN:
        lhs[0:idr, 0:ndr, :] = rhsr_b[0:idr, 1:ndr + 1, None]
        lhs[0, idr:idr + idl - 1, :] = lhs[0, idr:idr + idl - 1, :] - lam * rhs_b[0, idl + 1:ndl + 1, None]
        for ir in range(1, idr):
            lhs[ir, idr - idl:idr + idl - 1, :] = lhs[ir, idr - idl:idr + idl - 1, :] + lam * rhs_b[ir, 1:ndl + 1, None]
    else:
        lhs[nx - idr:nx, 0:ndr, :] = rhsr_t[1:idr + 1, 0:ndr, None]
        lhs[nx - 1, idr - idl:idr - 1, :] = lhs[nx - 1, idr - idl:idr - 1, :] - lam * rhs_t[idl, 0:idl - 1, None]
        for ir in range(1, idr):
            lhs[nx - ir - 1, idr - idl:idr + idl - 1, :] = lhs[nx - ir - 1, idr - idl:idr + idl - 1, :] + lam * rhs_t[idl - ir, 0:ndl, None]
    # normalisation (:175-201)
    rhs = A
    mx = max(idr, idl + 1)
    for ir in range(1, mx + 1):
        dummy = 1.0 / rhs[ir - 1, idl - 1]
        rhs_b[ir - 1, 0:ndl + 1] = rhs_b[ir - 1, 0:ndl + 1] * dummy
        dummy = 1.0 / rhs[nx - ir, idl - 1]
        rhs_t[idl - ir + 1, 0:ndl + 1] = rhs_t[idl - ir + 1, 0:ndl + 1] * dummy
        dummy = 1.0 / rhs[ir - 1, idl - 1]
        rhs[ir - 1, 0:ndl] = rhs[ir - 1, 0:ndl] * dummy
        lhs[ir - 1, 0:ndr, :] = lhs[ir - 1, 0:ndr, :] * dummy
        dummy = 1.0 / rhs[nx - ir, idl - 1]
        rhs[nx - ir, 0:ndl] = rhs[nx - ir, 0:ndl] * dummy
        lhs[nx - ir, 0:ndr, :] = lhs[nx - ir, 0:ndr, :] * dummy
    for ir in range(mx + 1, nx - mx + 1):
        dummy = 1.0 / rhs[ir - 1, idl]
        rhs[ir - 1, 0:ndl] = rhs[ir - 1, 0:ndl] * dummy
        lhs[ir - 1, 0:ndr, :] = lhs[ir - 1, 0:ndr, :] * dummy
    # reduce the opposite end (:203-211); makes rhs_b / rhs_t lambda-dependent
    rhs_bM = np.repeat(rhs_b[:, :, None], M, axis=2)
    rhs_tM = np.repeat(rhs_t[:, :, None], M, axis=2)
    rhsM = np.repeat(rhs[:, :, None], M, axis=2)
    if ibc == BCS_MIN:
        fdm_bcs_reduce(BCS_MAX, lhs, rhsM, rhs_t=rhs_tM)
    else:
        fdm_bcs_reduce(BCS_MIN, lhs, rhsM, rhs_b=rhs_bM)
    p.lhs, p.rhs, p.rhs_b, p.rhs_t = lhs, rhs, rhs_bM, rhs_tM
    p.factorized = False
    return p


def int1_initialize(g, lam, ibc):
    """fdm/fdm_integral.f90:58-87 FDM_Int1_Initialize: TRIDFS / PENTADFS / HEPTADFS by the number of diagonals of the integral system
    (= RHS diagonals of the derivative: 3 for CompactJacobian4 / CompactDirect4, 5 for CompactJacobian6, 7 for CompactJacobian6Penta)."""
    p = int1_create_system(g, lam, ibc)
    nd = p.lhs.shape[1]
    cols = [p.lhs[1:p.nx - 1, k, :].copy() for k in range(nd)]
    {3: tridfs_m, 5: pentadfs, 7: heptadfs}[nd](*cols)
    for k in range(nd):
        p.lhs[1:p.nx - 1, k, :] = cols[k]
    p.factorized = True
    return p


def _matmul_3d_both(rhs, f, res, rhs_b, rhs_t):
    """fdm/fdm_matmul.f90:70-121 MatMul_3d with ibc = BCS_BOTH, as called at fdm_integral.f90:249-250:
    rhs_b = fdmi%rhs_b(1:3, 0:3), rhs_t = fdmi%rhs_t(0:2, 1:4).  res[0], res[nx-1] carry the boundary values.
    Returns (bcs_b, bcs_t)."""
    nx = rhs.shape[0]
    r1, r2 = rhs[:, 0], rhs[:, 1]
    rb = lambda j, c: rhs_b[j - 1, c]          # noqa: E731  rhs_b(j, c)
    rt = lambda r, c: rhs_t[r, c - 1]          # noqa: E731  rhs_t(r, c)
    bcs_b = res[0] * rb(1, 2) + f[1] * rb(1, 3) + f[2] * rb(1, 1)
    res[1] = res[0] * rb(2, 1) + f[1] * rb(2, 2) + f[2] * rb(2, 3)
    res[2] = res[0] * rb(3, 0) + f[1] * rb(3, 1) + f[2] * rb(3, 2) + f[3] * rb(3, 3)
    for n in range(3, nx - 3):
        res[n] = f[n - 1] * r1[n] + f[n] * r2[n] + f[n + 1]
    res[nx - 3] = f[nx - 4] * rt(0, 1) + f[nx - 3] * rt(0, 2) + f[nx - 2] * rt(0, 3) + res[nx - 1] * rt(0, 4)
    res[nx - 2] = f[nx - 3] * rt(1, 1) + f[nx - 2] * rt(1, 2) + res[nx - 1] * rt(1, 3)
    bcs_t = f[nx - 3] * rt(2, 3) + f[nx - 2] * rt(2, 1) + res[nx - 1] * rt(2, 2)
    return bcs_b, bcs_t


def _matmul_5d_both(rhs, f, res, rhs_b, rhs_t):
    """fdm/fdm_matmul.f90:267-320 MatMul_5d with ibc = BCS_BOTH, as called at fdm_integral.f90:251-253: rhs_b = fdmi%rhs_b(1:4, 0:5),
    rhs_t = fdmi%rhs_t(0:3, 1:6).  res[0], res[nx-1] carry the boundary values.  Returns (bcs_b, bcs_t)."""
    nx = rhs.shape[0]
    r1, r2, r3, r5 = rhs[:, 0], rhs[:, 1], rhs[:, 2], rhs[:, 4]
    rb = lambda j, c: rhs_b[j - 1, c]          # noqa: E731  rhs_b(j, c)
    rt = lambda r, c: rhs_t[r, c - 1]          # noqa: E731  rhs_t(r, c)
    bcs_b = res[0] * rb(1, 3) + f[1] * rb(1, 4) + f[2] * rb(1, 5) + f[3] * rb(1, 1)
    res[1] = res[0] * rb(2, 2) + f[1] * rb(2, 3) + f[2] * rb(2, 4) + f[3] * rb(2, 5)
    res[2] = res[0] * rb(3, 1) + f[1] * rb(3, 2) + f[2] * rb(3, 3) + f[3] * rb(3, 4) + f[4] * rb(3, 5)
    res[3] = res[0] * rb(4, 0) + f[1] * rb(4, 1) + f[2] * rb(4, 2) + f[3] * rb(4, 3) + f[4] * rb(4, 4) + f[5] * rb(4, 5)
    for n in range(4, nx - 4):
        res[n] = f[n - 2] * r1[n] + f[n - 1] * r2[n] + f[n] * r3[n] + f[n + 1] + f[n + 2] * r5[n]
    res[nx - 4] = f[nx - 6] * rt(0, 1) + f[nx - 5] * rt(0, 2) + f[nx - 4] * rt(0, 3) + f[nx - 3] * rt(0, 4) + f[nx - 2] * rt(0, 5) + res[nx - 1] * rt(0, 6)
    res[nx - 3] = f[nx - 5] * rt(1, 1) + f[nx - 4] * rt(1, 2) + f[nx - 3] * rt(1, 3) + f[nx - 2] * rt(1, 4) + res[nx - 1] * rt(1, 5)
    res[nx - 2] = f[nx - 4] * rt(2, 1) + f[nx - 3] * rt(2, 2) + f[nx - 2] * rt(2, 3) + res[nx - 1] * rt(2, 4)
    bcs_t = f[nx - 4] * rt(3, 5) + f[nx - 3] * rt(3, 1) + f[nx - 2] * rt(3, 2) + res[nx - 1] * rt(3, 3)
    return bcs_b, bcs_t


def int1_solve(p, rhsi, f, res, want_du=False):
    """fdm/fdm_integral.f90:219-314 FDM_Int1_Solve.  f, res: (n, nlines, M); res carries the boundary value
    (res[0] for BCS_MIN, res[nx-1] for BCS_MAX) and is overwritten with the solution.  Returns du_boundary or None."""
    nx = p.nx
    lhs = p.lhs
    if p.bc == BCS_MIN:
        res[nx - 1] = f[nx - 1]
    else:
        res[0] = f[0]
    ndl, ndr = lhs.shape[1], rhsi.shape[1]            # diagonals of the integral system / of its right-hand side (fdm_integral.f90:236-239)
    idl, idr = ndl // 2 + 1, ndr // 2 + 1
    bcs_b, bcs_t = (_matmul_3d_both if ndr == 3 else _matmul_5d_both)(rhsi, f, res, p.rhs_b, p.rhs_t)
    sub = res[1:nx - 1]
    {3: tridss_m, 5: pentadss, 7: heptadss}[ndl](*([lhs[1:nx - 1, k] for k in range(ndl)] + [sub]))
    du = None
    if p.bc == BCS_MAX:
        res[0] = bcs_b
        for ic in range(1, idl):
            res[0] = res[0] + lhs[0, idl + ic - 1] * res[ic]
        res[0] = res[0] + lhs[0, 0] * res[idl]
        if want_du:
            du = lhs[nx - 1, idl - 1] * res[nx - 1]
            for ic in range(1, idl):
                du = du + lhs[nx - 1, idl - ic - 1] * res[nx - 1 - ic]
            du = du + lhs[nx - 1, ndl - 1] * res[nx - 1 - idl]
            for ic in range(1, idr):
                du = du + rhsi[nx - 1, idr - ic - 1] * f[nx - 1 - ic]
    if p.bc == BCS_MIN:
        res[nx - 1] = bcs_t
        for ic in range(1, idl):
            res[nx - 1] = res[nx - 1] + lhs[nx - 1, idl - ic - 1] * res[nx - 1 - ic]
        res[nx - 1] = res[nx - 1] + lhs[nx - 1, ndl - 1] * res[nx - 1 - idl]
        if want_du:
            du = lhs[0, idl - 1] * res[0]
            for ic in range(1, idl):
                du = du + lhs[0, idl + ic - 1] * res[ic]
            du = du + lhs[0, 0] * res[idl]
            for ic in range(1, idr):
                du = du + rhsi[0, idr + ic - 1] * f[ic]
    return du


# ######################################################################################
# operators/opr_odes.f90
# ######################################################################################
def ode2_factorize_nn(fmin, fmax, f, bcs):
    """operators/opr_odes.f90:265-386 OPR_ODE2_Factorize_NN.  fmin/fmax: Int1 plans (BCS_MIN, +lambda) / (BCS_MAX, -lambda);
    f: (n, nlines, M) (modified like in the reference), bcs: (2, nlines, M) [bottom, top].  Returns (u, v)."""
    lam = fmin.lam
    nx = fmin.nx
    nl, M = f.shape[1], f.shape[2]
    u = np.zeros_like(f)
    v = np.zeros_like(f)
    # v^(0): v' + lambda v = f, v_1 = 0
    f[nx - 1] = 0.0
    v[0] = 0.0
    int1_solve(fmin, fmin.rhs, f, v)
    # v^(1), e^(-) (third line unused)
    f1 = np.zeros((nx, 3, M))
    h2 = np.zeros((nx, 3, M))          # (v1, em, dd)
    f1[nx - 1, 0] = 1.0
    h2[0, 0] = 0.0
    h2[0, 1] = 1.0
    h2[0, 2] = 0.0
    int1_solve(fmin, fmin.rhs, f1, h2)
    # u^(0): u' - lambda u = v, u_n = 0
    u[nx - 1] = 0.0
    du0_n = int1_solve(fmax, fmax.rhs, v, u, want_du=True)
    # u^(1), s^(+), e^(+)
    h1 = np.zeros((nx, 3, M))          # (u1, sp, ep)
    h2[:, 2] = 0.0
    h1[nx - 1, 0] = 0.0
    h1[nx - 1, 1] = 0.0
    h1[nx - 1, 2] = 1.0
    der_bcs = int1_solve(fmax, fmax.rhs, h2, h1, want_du=True)
    v1, em = h2[:, 0], h2[:, 1]
    u1, sp, ep = h1[:, 0], h1[:, 1], h1[:, 2]
    du1_n, dsp_n, dep_n = der_bcs[0], der_bcs[1], der_bcs[2]
    a11 = 1.0 + lam * sp[0]; a21 = em[nx - 1]; a31 = dsp_n
    a12 = lam * ep[0]; a22 = lam * np.ones(M); a32 = dep_n
    a13 = lam * u1[0]; a23 = v1[nx - 1]; a33 = du1_n
    a12 = a12 / a11
    a22 = a22 - a21 * a12
    a32 = a32 - a31 * a12
    a13 = a13 / a11
    a23 = (a23 - a21 * a13) / a22
    a33 = a33 - a31 * a13 - a32 * a23
    v[0] = (bcs[0] - lam * u[0]) / a11
    u[nx - 1] = (bcs[1] - v[nx - 1] - a21 * v[0]) / a22
    fn = (bcs[1] - du0_n - a31 * v[0] - a32 * u[nx - 1]) / a33
    u[nx - 1] = u[nx - 1] - a23 * fn
    v[0] = v[0] - a12 * u[nx - 1] - a13 * fn
    i = nx - 1
    v[i] = v[i] + fn * v1[i] + v[0] * em[i] + lam * u[i]
    for i in range(nx - 2, 0, -1):
        u[i] = u[i] + fn * u1[i] + v[0] * sp[i] + u[nx - 1] * ep[i]
        v[i] = v[i] + fn * v1[i] + v[0] * em[i] + lam * u[i]
    i = 0
    u[i] = u[i] + fn * u1[i] + v[0] * sp[i] + u[nx - 1] * ep[i]
    v[i] = v[i] + lam * u[i]
    return u, v


def ode2_factorize_dn_sing(fmin, fmax, f, bcs):
    """operators/opr_odes.f90:37-96 OPR_ODE2_Factorize_DN_Sing (lambda = 0 plans)."""
    nx = fmin.nx
    M = f.shape[2]
    u = np.zeros_like(f)
    v = np.zeros_like(f)
    f[0] = 0.0
    v[nx - 1] = bcs[1]
    int1_solve(fmax, fmax.rhs, f, v)
    f1 = np.zeros((nx, 1, M)); f1[0, 0] = 1.0
    v1 = np.zeros((nx, 1, M))
    int1_solve(fmax, fmax.rhs, f1, v1)
    u[0] = bcs[0]
    du0_n = int1_solve(fmin, fmin.rhs, v, u, want_du=True)
    u1 = np.zeros((nx, 1, M))
    du1_n = int1_solve(fmin, fmin.rhs, v1, u1, want_du=True)
    fac = 1.0 / (du1_n[0] - v1[0, 0])
    c = (v[0] - du0_n) * fac
    for i in range(nx):
        u[i] = u[i] + c * u1[i, 0]
        v[i] = v[i] + c * v1[i, 0]
    return u, v


def ode2_factorize_nn_sing(fmin, fmax, f, bcs):
    """operators/opr_odes.f90:165-183 OPR_ODE2_Factorize_NN_Sing."""
    bcs = bcs.copy()
    bcs[0] = 0.0
    return ode2_factorize_dn_sing(fmin, fmax, f, bcs)


def ode2_factorize_dd(fmin, fmax, f, bcs):
    """operators/opr_odes.f90:391-478 OPR_ODE2_Factorize_DD: u given at both ends.  Same conventions as ode2_factorize_nn."""
    lam = fmin.lam
    nx = fmin.nx
    M = f.shape[2]
    u = np.zeros_like(f)
    v = np.zeros_like(f)
    f[nx - 1] = 0.0                                     # v^(0): v' + lambda v = f, v_1 = 0
    v[0] = 0.0
    int1_solve(fmin, fmin.rhs, f, v)
    f1 = np.zeros((nx, 2, M))                           # v^(1), e^(-)
    h2 = np.zeros((nx, 2, M))
    f1[nx - 1, 0] = 1.0
    h2[0, 0] = 0.0
    h2[0, 1] = 1.0
    int1_solve(fmin, fmin.rhs, f1, h2)
    u[nx - 1] = bcs[1]                                  # u^(0): u' - lambda u = v, u_n given
    du0_n = int1_solve(fmax, fmax.rhs, v, u, want_du=True)
    h1 = np.zeros((nx, 2, M))                           # u^(1), s^(+)
    der_bcs = int1_solve(fmax, fmax.rhs, h2, h1, want_du=True)
    v1, em = h2[:, 0], h2[:, 1]
    u1, sp = h1[:, 0], h1[:, 1]
    du1_n, dsp_n = der_bcs[0], der_bcs[1]
    aa = du1_n - v1[nx - 1]
    bb = dsp_n - em[nx - 1]
    dummy = 1.0 / (aa * sp[0] - bb * u1[0])
    q1 = (aa * (bcs[0] - u[0]) - u1[0] * (lam * bcs[1] - du0_n + v[nx - 1])) * dummy
    fn = (sp[0] * (lam * bcs[1] - du0_n + v[nx - 1]) - bb * (bcs[0] - u[0])) * dummy
    v[0] = q1
    for i in range(nx - 1, 0, -1):
        u[i] = u[i] + fn * u1[i] + v[0] * sp[i]
        v[i] = v[i] + fn * v1[i] + v[0] * em[i] + lam * u[i]
    u[0] = bcs[0]
    v[0] = v[0] + lam * u[0]
    return u, v


def ode2_factorize_dd_sing(fmin, fmax, f, bcs):
    """operators/opr_odes.f90:188-260 OPR_ODE2_Factorize_DD_Sing (lambda = 0 plans)."""
    nx = fmin.nx
    M = f.shape[2]
    u = np.zeros_like(f)
    v = np.zeros_like(f)
    f[nx - 1] = 0.0                                     # v^(0): v' = f, v_1 = 0
    v[0] = 0.0
    int1_solve(fmin, fmin.rhs, f, v)
    f1 = np.zeros((nx, 1, M)); f1[nx - 1, 0] = 1.0      # v^(1)
    v1 = np.zeros((nx, 1, M))
    int1_solve(fmin, fmin.rhs, f1, v1)
    u[nx - 1] = bcs[1]                                  # u^(0): u' = v, u_n given
    du0_n = int1_solve(fmax, fmax.rhs, v, u, want_du=True)
    u1 = np.zeros((nx, 1, M))                           # u^(1)
    du1_n = int1_solve(fmax, fmax.rhs, v1, u1, want_du=True)
    f1 = np.ones((nx, 1, M))                            # s^(+) = x - x_n
    sp = np.zeros((nx, 1, M))
    int1_solve(fmax, fmax.rhs, f1, sp)
    fn = 1.0 / (du1_n[0] - v1[nx - 1, 0])
    c = (v[nx - 1] - du0_n) * fn
    dummy = 1.0 / sp[0, 0]
    v[0] = (bcs[0] - (u[0] + c * u1[0, 0])) * dummy
    u[0] = bcs[0]
    for i in range(1, nx):
        u[i] = u[i] + c * u1[i, 0] + v[0] * sp[i, 0]
        v[i] = v[i] + c * v1[i, 0] + v[0]
    return u, v


# ######################################################################################
# operators/opr_elliptic.f90
# ######################################################################################
class PoissonPlan:
    """operators/opr_elliptic.f90:86-250 OPR_Elliptic_Initialize (TYPE_FACTORIZE, serial): lambda(k,i), singular modes,
    norm; the integral plans are rebuilt per call in this oracle (they are cheap here)."""

    def __init__(self, gx, gy, gz, nx, ny, nz, stagger=False):
        self.nx, self.ny, self.nz = nx, ny, nz
        self.gy = gy
        self.nxh = nx // 2 + 1
        kx = gx.der1.mwn[: self.nxh]
        if nz > 1:
            kz = gz.der1.mwn[:nz]
            self.lam2 = kx[None, :] ** 2.0 + kz[:, None] ** 2.0          # lambda(k, i)  [kz, kx]
        else:
            self.lam2 = (kx[None, :] ** 2.0) * np.ones((1, 1))
        self.norm = 1.0 / float(nx * nz)
        self.i_sing = (0, nx // 2)                                       # 0-based (1, nx/2+1)
        self.k_sing = (0, nz // 2) if nz > 1 else (0, 0)
        if stagger:                                                      # only one singular mode + other modified wavenumbers (:144-146)
            self.i_sing, self.k_sing = (0, 0), (0, 0)
        sing = np.zeros((max(nz, 1), self.nxh), dtype=bool)
        for i in set(self.i_sing):
            for k in set(self.k_sing):
                sing[k, i] = True
        self.sing = sing


def opr_poisson_fxz(plan, p, bcs_hb, bcs_ht, ibc=BCS_NN):
    """operators/opr_elliptic.f90:263-364 OPR_Poisson_FourierXZ_Factorize (ibc = BCS_NN).
    p: flat forcing (nx*ny*nz, x fastest); bcs_hb, bcs_ht: (nz, nx) Neumann data.  Returns (p, dpdy) flat."""
    assert ibc in (BCS_NN, BCS_DD), "oracle: BCS_NN (the RHS call, rhs_global_incompressible_1.f90:284) and BCS_DD are restated"
    nx, ny, nz, nxh = plan.nx, plan.ny, plan.nz, plan.nxh
    a = np.array(p, dtype=np.float64).reshape(nz, ny, nx).copy()
    a[:, 0, :] = bcs_hb.reshape(nz, nx)                                  # :285-286
    a[:, ny - 1, :] = bcs_ht.reshape(nz, nx)
    c = np.fft.rfft(a, axis=2)                                           # OPR_Fourier_X_Forward (unnormalised, like FFTW)
    if nz > 1:
        c = np.fft.fft(c, axis=0)                                        # OPR_Fourier_Z_Forward
    c = c * plan.norm                                                    # :295
    # modes as a flat axis M = (kz, kx); lines = (Re, Im)
    M = nz * nxh
    f = np.empty((ny, 2, M))
    f[:, 0, :] = c.real.transpose(1, 0, 2).reshape(ny, M)
    f[:, 1, :] = c.imag.transpose(1, 0, 2).reshape(ny, M)
    bcs = np.stack([f[0].copy(), f[ny - 1].copy()])                      # :310-311
    lam = np.sqrt(plan.lam2.reshape(M))
    sing = plan.sing.reshape(M)
    u = np.zeros_like(f)
    v = np.zeros_like(f)
    reg = ~sing
    g1 = plan.gy.der1
    if reg.any():
        fmin = int1_initialize(g1, lam[reg], BCS_MIN)                   # opr_elliptic.f90:205-209
        fmax = int1_initialize(g1, -lam[reg], BCS_MAX)
        solve = ode2_factorize_nn if ibc == BCS_NN else ode2_factorize_dd                  # :315-329
        u[:, :, reg], v[:, :, reg] = solve(fmin, fmax, f[:, :, reg].copy(), bcs[:, :, reg])
    if sing.any():
        ls = lam[sing]                                                   # exactly 0 at (0|Nyquist) x (0|Nyquist)
        fmin = int1_initialize(g1, ls, BCS_MIN)
        fmax = int1_initialize(g1, -ls, BCS_MAX)
        solve = ode2_factorize_nn_sing if ibc == BCS_NN else ode2_factorize_dd_sing
        u[:, :, sing], v[:, :, sing] = solve(fmin, fmax, f[:, :, sing].copy(), bcs[:, :, sing])

    def back(w):
        cc = (w[:, 0, :] + 1j * w[:, 1, :]).reshape(ny, nz, nxh).transpose(1, 0, 2)
        if nz > 1:
            cc = np.fft.ifft(cc, axis=0) * nz                            # FFTW backward is unnormalised
        return (np.fft.irfft(cc, n=nx, axis=2) * nx).reshape(-1)

    return back(u), back(v)


def opr_helmholtz_fxz_factorize(plan, a, bcs_hb, bcs_ht, ibc, alpha):
    """operators/opr_elliptic.f90:466-557 OPR_Helmholtz_FourierXZ_Factorize: lap a + alpha a = f.  Per mode the two first-order systems of
    OPR_ODE2_Factorize_NN / _DD with sqrt(lambda(k,i) - alpha) (:518-522), every mode a regular one (no singular-mode branch).  Returns a flat."""
    assert ibc in (BCS_NN, BCS_DD)                                       # :524-532
    nx, ny, nz, nxh = plan.nx, plan.ny, plan.nz, plan.nxh
    w = np.array(a, dtype=np.float64).reshape(nz, ny, nx).copy()
    w[:, 0, :] = bcs_hb.reshape(nz, nx)                                  # :487-488
    w[:, ny - 1, :] = bcs_ht.reshape(nz, nx)
    c = np.fft.rfft(w, axis=2)
    if nz > 1:
        c = np.fft.fft(c, axis=0)
    c = c * plan.norm                                                    # :497
    M = nz * nxh
    f = np.empty((ny, 2, M))
    f[:, 0, :] = c.real.transpose(1, 0, 2).reshape(ny, M)
    f[:, 1, :] = c.imag.transpose(1, 0, 2).reshape(ny, M)
    bcs = np.stack([f[0].copy(), f[ny - 1].copy()])                      # :514-515
    lam = np.sqrt(plan.lam2.reshape(M) - alpha)
    fmin = int1_initialize(plan.gy.der1, lam, BCS_MIN)
    fmax = int1_initialize(plan.gy.der1, -lam, BCS_MAX)
    solve = ode2_factorize_nn if ibc == BCS_NN else ode2_factorize_dd
    u, _ = solve(fmin, fmax, f.copy(), bcs)
    cc = (u[:, 0, :] + 1j * u[:, 1, :]).reshape(ny, nz, nxh).transpose(1, 0, 2)
    if nz > 1:
        cc = np.fft.ifft(cc, axis=0) * nz
    return (np.fft.irfft(cc, n=nx, axis=2) * nx).reshape(-1)


# ######################################################################################
# DIRECT elliptic solver (EllipticOrder = CompactDirect4/6): fdm/fdm_integral.f90:318-673, operators/opr_elliptic.f90:368-455
# ######################################################################################
def _Pi(x, j, idx):
    """fdm/fdm_base.f90:31-44 (1-based j, idx)."""
    f = 1.0
    for k in idx:
        f = f * (x[j - 1] - x[k - 1])
    return f


def _Pi_p(x, j, idx):
    """fdm/fdm_base.f90:47-67."""
    f = 0.0
    for k in range(len(idx)):
        dummy = 1.0
        for m in range(len(idx)):
            if m != k:
                dummy = dummy * (x[j - 1] - x[idx[m] - 1])
        f = f + dummy
    return f


def _Pi_pp_3(x, j, idx):
    """fdm/fdm_base.f90:70-78."""
    return 2.0 * (x[j - 1] - x[idx[0] - 1] + x[j - 1] - x[idx[1] - 1] + x[j - 1] - x[idx[2] - 1])


def _Lag(x, j, i, idx):
    """fdm/fdm_base.f90:82-97."""
    f = 1.0
    for k in idx:
        if k != i:
            f = f * (x[j - 1] - x[k - 1]) / (x[i - 1] - x[k - 1])
    return f


def _Lag_p(x, j, i, idx):
    """fdm/fdm_base.f90:100-125."""
    den, f = 1.0, 0.0
    for k in range(len(idx)):
        if idx[k] != i:
            dummy = 1.0
            for m in range(len(idx)):
                if idx[m] != i and m != k:
                    dummy = dummy * (x[j - 1] - x[idx[m] - 1])
            f = f + dummy
            den = den * (x[i - 1] - x[idx[k] - 1])
    return f / den


def coef_c1n4_biased(x, i, backwards=False):
    """fdm/fdm_integral.f90:560-621 (contained in FDM_Int2_CreateSystem): p'_1 = b1 p1 + b2 p2 + b3 p3 + b4 p4 + a2 p''_2.  1-based i."""
    i1 = i
    i2, i3, i4 = (i - 1, i - 2, i - 3) if backwards else (i + 1, i + 2, i + 3)
    X = lambda k: x[k - 1]      # noqa: E731
    dx1, dx3, dx4 = X(i2) - X(i1), X(i2) - X(i3), X(i2) - X(i4)
    sm = [i1, i3, i4]
    a2 = 0.5 * (_Pi(x, i1, sm) - dx1 * _Pi_p(x, i1, sm)) / _Pi_p(x, i2, sm)
    b2 = _Pi_p(x, i1, sm) * (2.0 * _Pi_p(x, i2, sm) + dx1 * _Pi_pp_3(x, i2, sm)) - _Pi(x, i1, sm) * _Pi_pp_3(x, i2, sm)
    b2 = 0.5 * b2 / _Pi(x, i2, sm) / _Pi_p(x, i2, sm)

    def bk(ik, dxk):
        D = _Lag(x, i2, ik, sm) + dxk * _Lag_p(x, i2, ik, sm)
        b = _Lag(x, i1, ik, sm) * (_Lag(x, i2, ik, sm) + 2 * dx1 * _Lag_p(x, i2, ik, sm)) \
            - dx1 * _Lag_p(x, i1, ik, sm) * (_Lag(x, i2, ik, sm) + dx1 * _Lag_p(x, i2, ik, sm))
        return -b / dxk / D

    return np.array([bk(i1, dx1), b2, bk(i3, dx3), bk(i4, dx4), a2])


class Int2Plan:
    pass


def int2_create_system(g, x, lam2, ibc):
    """fdm/fdm_integral.f90:366-557 FDM_Int2_CreateSystem: (B - lambda2 A) u = A f with both boundary values given (Dirichlet), or the
    Neumann values folded in through a 4th-order biased formula.  g: oracle DerPlan of the second derivative (tables), x: nodes,
    lam2: (M,), ibc in BCS_DD/ND/DN/NN.  lhs: (n, ndr, M); rhs (n, ndl), rhs_b, rhs_t: lambda-independent."""
    lam = np.atleast_1d(np.asarray(lam2, dtype=np.float64))
    M = lam.shape[0]
    ndl, ndr = g.nb_diag
    idl, idr = ndl // 2 + 1, ndr // 2 + 1
    nx = g.size
    assert abs(idl - idr) <= 1
    p = Int2Plan()
    p.lam, p.bc, p.nx = lam, ibc, nx
    A = g.lhs[:, :ndl].copy()                                           # fdmi%rhs (:393)
    rhsr_b, rhsr_t = np.zeros((5, 8)), np.zeros((5, 8))
    fdm_bcs_reduce(BCS_BOTH, A, g.rhs[:, :ndr], rhsr_b, rhsr_t)         # :395
    rhs_b, rhs_t = np.zeros((5, 8)), np.zeros((5, 8))
    rhs_b[0:idl + 1, 1:ndl + 1] = A[0:idl + 1, 0:ndl]                   # :400-403
    for ir in range(1, idr):
        rhs_b[ir, idl - ir] = -rhsr_b[ir, idr - ir]
    rhs_t[0:idl + 1, 0:ndl] = A[nx - idl - 1:nx, 0:ndl]                 # :405-408
    for ir in range(1, idr):
        rhs_t[idl - ir, idl + ir - 1] = -rhsr_t[idr - ir, idr + ir - 1]
    # new lhs diagonals C = B - lambda2 A (:412-418)
    lhs = np.repeat(g.rhs[:, :ndr, None], M, axis=2).astype(np.float64)
    lhs[:, idr - 1, :] = lhs[:, idr - 1, :] - lam * g.lhs[:, idl - 1, None]
    for i in range(1, idl):
        lhs[i:nx, idr - i - 1, :] = lhs[i:nx, idr - i - 1, :] - lam * g.lhs[i:nx, idl - i - 1, None]
        lhs[0:nx - i, idr + i - 1, :] = lhs[0:nx - i, idr + i - 1, :] - lam * g.lhs[0:nx - i, idl + i - 1, None]
    lhs[1:idr, 0:ndr, :] = rhsr_b[1:idr, 1:ndr + 1, None]               # :422-425
    for ir in range(1, idr):
        lhs[ir, idr - idl:idr + idl - 1, :] = lhs[ir, idr - idl:idr + idl - 1, :] - lam * rhs_b[ir, 1:ndl + 1, None]
    lhs[nx - idr:nx - 1, 0:ndr, :] = rhsr_t[1:idr, 0:ndr, None]         # :429-432
    for ir in range(1, idr):
        lhs[nx - ir - 1, idr - idl:idr + idl - 1, :] = lhs[nx - ir - 1, idr - idl:idr + idl - 1, :] - lam * rhs_t[idl - ir, 0:ndl, None]
    # Neumann corrections (:436-514)
    if ibc in (BCS_ND, BCS_NN):
        coef = coef_c1n4_biased(x, 1)
        lhs[0, :, :] = 0.0
        lhs[0, 0:3, :] = (-coef[1:4] / coef[0])[:, None]
        rhs_b[0, :] = 0.0
        rhs_b[0, idl] = 1.0 / coef[0]
        rhs_b[0, idl + 1] = -coef[4] / coef[0]
        lhs[0, 0, :] = lhs[0, 0, :] + lam * rhs_b[0, idl + 1]
        for ir in range(1, idr):
            lhs[ir, idr - ir:idr - ir + 3, :] = lhs[ir, idr - ir:idr - ir + 3, :] - rhs_b[ir, idl - ir] * lhs[0, 0:3, :]
            rhs_b[ir, idl - ir + 1] = rhs_b[ir, idl - ir + 1] + rhs_b[ir, idl - ir] * rhs_b[0, idl + 1]
            rhs_b[ir, idl - ir] = rhs_b[ir, idl - ir] * rhs_b[0, idl]
    if ibc in (BCS_DN, BCS_NN):
        coef = coef_c1n4_biased(x, nx, backwards=True)
        lhs[nx - 1, :, :] = 0.0
        lhs[nx - 1, ndr - 3:ndr, :] = (-coef[[3, 2, 1]] / coef[0])[:, None]
        rhs_t[idl, :] = 0.0
        rhs_t[idl, idl - 1] = 1.0 / coef[0]
        rhs_t[idl, idl - 2] = -coef[4] / coef[0]
        lhs[nx - 1, ndr - 1, :] = lhs[nx - 1, ndr - 1, :] + lam * rhs_t[idl, idl - 2]
        for ir in range(1, idr):
            lhs[nx - ir - 1, ir - 1:ir + 2, :] = lhs[nx - ir - 1, ir - 1:ir + 2, :] - rhs_t[idl - ir, idl + ir - 1] * lhs[nx - 1, ndr - 3:ndr, :]
            rhs_t[idl - ir, idl + ir - 2] = rhs_t[idl - ir, idl + ir - 2] + rhs_t[idl - ir, idl + ir - 1] * rhs_t[idl, idl - 2]
            rhs_t[idl - ir, idl + ir - 1] = rhs_t[idl - ir, idl + ir - 1] * rhs_t[idl, idl - 1]
    # normalisation (:518-540): rows 2.. only
    rhs = A
    mx = max(idr, idl + 1)
    for ir in range(2, mx + 1):
        dummy = 1.0 / rhs[ir - 1, idl - 1]
        rhs_b[ir - 1, 0:ndl + 1] = rhs_b[ir - 1, 0:ndl + 1] * dummy
        dummy = 1.0 / rhs[nx - ir, idl - 1]
        rhs_t[idl - ir + 1, 0:ndl + 1] = rhs_t[idl - ir + 1, 0:ndl + 1] * dummy
        dummy = 1.0 / rhs[ir - 1, idl - 1]
        rhs[ir - 1, 0:ndl] = rhs[ir - 1, 0:ndl] * dummy
        lhs[ir - 1, 0:ndr, :] = lhs[ir - 1, 0:ndr, :] * dummy
        dummy = 1.0 / rhs[nx - ir, idl - 1]
        rhs[nx - ir, 0:ndl] = rhs[nx - ir, 0:ndl] * dummy
        lhs[nx - ir, 0:ndr, :] = lhs[nx - ir, 0:ndr, :] * dummy
    for ir in range(mx + 1, nx - mx + 1):
        dummy = 1.0 / rhs[ir - 1, idl]
        rhs[ir - 1, 0:ndl] = rhs[ir - 1, 0:ndl] * dummy
        lhs[ir - 1, 0:ndr, :] = lhs[ir - 1, 0:ndr, :] * dummy
    p.lhs, p.rhs, p.rhs_b, p.rhs_t = lhs, rhs, rhs_b, rhs_t
    p.factorized = False
    return p


def int2_initialize(g, x, lam2, ibc):
    """fdm/fdm_integral.f90:334-361 FDM_Int2_Initialize (5 LHS diagonals: PENTADFS on rows 2..n-1)."""
    p = int2_create_system(g, x, lam2, ibc)
    assert p.lhs.shape[1] == 5, "oracle: only the pentadiagonal second-order integral (tridiagonal A, pentadiagonal B) is restated"
    cols = [p.lhs[1:p.nx - 1, k, :].copy() for k in range(5)]
    pentadfs(*cols)
    for k in range(5):
        p.lhs[1:p.nx - 1, k, :] = cols[k]
    p.factorized = True
    return p


def int2_solve(p, rhsi, f, res):
    """fdm/fdm_integral.f90:626-671 FDM_Int2_Solve.  f, res: (n, nlines, M); res[0], res[n-1] carry the boundary values (function value for
    a Dirichlet end, derivative for a Neumann end); res is overwritten with the solution."""
    nx, lhs = p.nx, p.lhs
    assert rhsi.shape[1] == 3
    bcs_b, bcs_t = _matmul_3d_both(rhsi, f, res, p.rhs_b, p.rhs_t)
    pentadss(lhs[1:nx - 1, 0], lhs[1:nx - 1, 1], lhs[1:nx - 1, 2], lhs[1:nx - 1, 3], lhs[1:nx - 1, 4], res[1:nx - 1])
    ndl = 5
    if p.bc in (BCS_ND, BCS_NN):
        res[0] = bcs_b + lhs[0, 0] * res[1] + lhs[0, 1] * res[2] + lhs[0, 2] * res[3]
    if p.bc in (BCS_DN, BCS_NN):
        res[nx - 1] = bcs_t + lhs[nx - 1, ndl - 1] * res[nx - 2] + lhs[nx - 1, ndl - 2] * res[nx - 3] + lhs[nx - 1, ndl - 3] * res[nx - 4]
    return res


class PoissonDirectPlan:
    """operators/opr_elliptic.f90:152-163,228-245 OPR_Elliptic_Initialize (TYPE_DIRECT, serial): lambda(k,i) from the SECOND-derivative
    modified wavenumbers of x and z, one singular mode (1,1) solved with BCS_DN.  gy.der2 must hold a direct scheme (tables)."""

    def __init__(self, gx, gy, gz, nx, ny, nz):
        self.nx, self.ny, self.nz, self.nxh = nx, ny, nz, nx // 2 + 1
        self.gy = gy
        kx = gx.der2.mwn[: self.nxh]
        if nz > 1:
            self.lam2 = kx[None, :] + gz.der2.mwn[:nz][:, None]
        else:
            self.lam2 = kx[None, :] * np.ones((1, 1))
        self.norm = 1.0 / float(nx * nz)
        self.sing = np.zeros((max(nz, 1), self.nxh), dtype=bool)
        self.sing[0, 0] = True


def opr_poisson_fxz_direct(plan, p, bcs_hb, bcs_ht, ibc=BCS_NN, gy_der=None):
    """operators/opr_elliptic.f90:368-455 OPR_Poisson_FourierXZ_Direct.  Returns (p, dpdy) flat; dpdy = OPR_Partial_Y(OPR_P1) of p with the
    y plan of the derivatives (gy_der, default plan.gy), :447-449."""
    from .tlab_oracle import opr_partial
    nx, ny, nz, nxh = plan.nx, plan.ny, plan.nz, plan.nxh
    a = np.array(p, dtype=np.float64).reshape(nz, ny, nx).copy()
    a[:, 0, :] = bcs_hb.reshape(nz, nx)                                  # :392-393
    a[:, ny - 1, :] = bcs_ht.reshape(nz, nx)
    c = np.fft.rfft(a, axis=2)
    if nz > 1:
        c = np.fft.fft(c, axis=0)
    c = c * plan.norm                                                    # :402
    M = nz * nxh
    f = np.empty((ny, 2, M))
    f[:, 0, :] = c.real.transpose(1, 0, 2).reshape(ny, M)
    f[:, 1, :] = c.imag.transpose(1, 0, 2).reshape(ny, M)
    u = np.zeros_like(f)
    u[0], u[ny - 1] = f[0], f[ny - 1]                                    # :416-417
    lam = plan.lam2.reshape(M)
    sing = plan.sing.reshape(M)
    g2, x = plan.gy.der2, plan.gy.nodes
    reg = ~sing if ibc == BCS_NN else np.ones(M, dtype=bool)
    pr = int2_initialize(g2, x, lam[reg], ibc)
    ur = u[:, :, reg]
    int2_solve(pr, pr.rhs, f[:, :, reg], ur)
    u[:, :, reg] = ur
    if ibc == BCS_NN:                                                    # :420-424 with :236-240: compatibility constraint, p = 0 at the bottom
        ps = int2_initialize(g2, x, lam[sing], BCS_DN)
        us = u[:, :, sing]
        us[0] = 0.0
        int2_solve(ps, ps.rhs, f[:, :, sing], us)
        u[:, :, sing] = us
    cc = (u[:, 0, :] + 1j * u[:, 1, :]).reshape(ny, nz, nxh).transpose(1, 0, 2)
    if nz > 1:
        cc = np.fft.ifft(cc, axis=0) * nz
    pout = (np.fft.irfft(cc, n=nx, axis=2) * nx).reshape(-1)
    dpdy = opr_partial(2, 1, nx, ny, nz, 0, gy_der if gy_der is not None else plan.gy, pout)[0]
    return pout, dpdy


def opr_helmholtz_fxz_direct(plan, a, bcs_hb, bcs_ht, ibc, alpha):
    """operators/opr_elliptic.f90:562-628 OPR_Helmholtz_FourierXZ_Direct: lap a + alpha a = f; per mode FDM_Int2 with the constant
    lambda(k,i) - alpha and the boundary type ibc, no singular-mode treatment.  Returns a flat."""
    nx, ny, nz, nxh = plan.nx, plan.ny, plan.nz, plan.nxh
    w = np.array(a, dtype=np.float64).reshape(nz, ny, nx).copy()
    w[:, 0, :] = bcs_hb.reshape(nz, nx)                                  # :581-582
    w[:, ny - 1, :] = bcs_ht.reshape(nz, nx)
    c = np.fft.rfft(w, axis=2)
    if nz > 1:
        c = np.fft.fft(c, axis=0)
    c = c * plan.norm                                                    # :591
    M = nz * nxh
    f = np.empty((ny, 2, M))
    f[:, 0, :] = c.real.transpose(1, 0, 2).reshape(ny, M)
    f[:, 1, :] = c.imag.transpose(1, 0, 2).reshape(ny, M)
    u = np.zeros_like(f)
    u[0], u[ny - 1] = f[0], f[ny - 1]                                    # :601-602
    pr = int2_initialize(plan.gy.der2, plan.gy.nodes, plan.lam2.reshape(M) - alpha, ibc)     # :604
    int2_solve(pr, pr.rhs, f, u)
    cc = (u[:, 0, :] + 1j * u[:, 1, :]).reshape(ny, nz, nxh).transpose(1, 0, 2)
    if nz > 1:
        cc = np.fft.ifft(cc, axis=0) * nz
    return (np.fft.irfft(cc, n=nx, axis=2) * nx).reshape(-1)
