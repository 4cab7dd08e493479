"""CPU oracle: numpy restatement of Tlab's Navier-Stokes RHS hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path (tlab_amd) never does.

Parity status: PINNED.  Every function here is checked in tests/test_oracle_vs_ref.py against
  (a) oracle/_ref/libtlab_ref.so = the reference's own Fortran sources compiled in place (when present), and
  (b) the golden vectors in tests/golden/ that were generated from (a) by tests/golden/make_golden.py.

Each function cites the reference file:line (relative to /root/reference/src) it restates and follows the
reference's operation order so that results agree to a few ulp (flang -O2 contracts some a*b+c into FMAs,
so agreement is ~1e-15 relative, not always bitwise).

Array conventions: Fortran u(nlines, n) (lines fastest) is a C-ordered numpy array of shape (n, nlines).
3-D fields are flat arrays of length nx*ny*nz with x fastest, i.e. C-ordered shape (nz, ny, nx).
Coefficient tables lhs(n, ndl), rhs(n, ndr), lu(n, ncol) keep the reference's (row, diagonal) indexing with
0-based numpy indices: Fortran rhs(i, k) == rhs[i-1, k-1].  rhs_b(4, 0:7) -> array (4, 8) indexed
[row-1, col]; rhs_t(0:4, 7) -> array (5, 7) indexed [row, col-1].
"""
import numpy as np

# base/tlab_constants.f90:62-71
BCS_PERIODIC = -1
BCS_DD, BCS_ND, BCS_DN, BCS_NN = 0, 1, 2, 3
BCS_NONE, BCS_MIN, BCS_MAX, BCS_BOTH = 0, 1, 2, 3
PI = 3.14159265358979323846

# fdm/fdm_derivative.f90:51-58
FDM_COM4_JACOBIAN = 4
FDM_COM6_JACOBIAN_PENTA = 5
FDM_COM6_JACOBIAN = 6
FDM_COM6_JACOBIAN_HYPER = 7
FDM_COM6_DIRECT = 16
FDM_COM4_DIRECT = 17

# operators/opr_partial.f90:19-21 ; physics/opr_burgers.f90:29-30
OPR_P1, OPR_P2, OPR_P2_P1 = 1, 2, 3
OPR_P1_INT_VP, OPR_P1_INT_PV, OPR_P0_INT_VP, OPR_P0_INT_PV = 5, 6, 7, 8      # interpolatory operators of the staggered pressure grid (:22-25)
OPR_B_SELF, OPR_B_U_IN = 0, 1

# value the flang-built reference picks up for the out-of-bounds coef_bc1(7) of the C2N6-Hyper wall closure
# (see fdm_c2n6_hyper_jacobian below); part of the parity definition, documented in DESIGN.md
HYPER_BC1_EXT = 0.1
# the same defect in the first-derivative closure of CompactJacobian6Penta (fdm_com1_jacobian.f90:237): coef_bc1(7) reads coef_bc2(1) = 1/6
PENTA_BC1_EXT = 1.0 / 6.0


# ######################################################################################
# utils/linear3.f90 -- Thomas algorithm, LU stored the reference's way
# ######################################################################################
def tridfs(a, b, c):
    """utils/linear3.f90:29-51 TRIDFS. In place. a,b,c: (nmax,)"""
    nmax = a.shape[0]
    for n in range(1, nmax):
        a[n] = a[n] / b[n - 1]
        b[n] = b[n] - a[n] * c[n - 1]
    a[:] = -a
    b[:] = 1.0 / b
    c[:] = -c


def tridss(a, b, c, f):
    """utils/linear3.f90:56-150 TRIDSS. f: (nmax, len) in place."""
    nmax = a.shape[0]
    for n in range(1, nmax):
        f[n] = f[n] + a[n] * f[n - 1]
    f[nmax - 1] = f[nmax - 1] * b[nmax - 1]
    for n in range(nmax - 2, -1, -1):
        f[n] = (f[n] + c[n] * f[n + 1]) * b[n]


def tridpfs(a, b, c, d, e):
    """utils/linear3.f90:269-316 TRIDPFS (circulant LU). In place; d,e are outputs."""
    nmax = a.shape[0]
    c[0] = c[0] / b[0]
    e[0] = a[0] / b[0]
    d[0] = c[nmax - 1]
    for n in range(1, nmax - 2):
        b[n] = b[n] - a[n] * c[n - 1]
        c[n] = c[n] / b[n]
        e[n] = -a[n] * e[n - 1] / b[n]
        d[n] = -d[n - 1] * c[n - 1]
    b[nmax - 2] = b[nmax - 2] - a[nmax - 2] * c[nmax - 3]
    e[nmax - 2] = (c[nmax - 2] - a[nmax - 2] * e[nmax - 3]) / b[nmax - 2]
    d[nmax - 2] = a[nmax - 1] - d[nmax - 3] * c[nmax - 3]
    s = 0.0
    for n in range(nmax - 1):
        s = s + d[n] * e[n]
    b[nmax - 1] = b[nmax - 1] - s
    for n in range(nmax):
        b[n] = 1.0 / b[n]
        a[n] = -a[n] * b[n]
        c[n] = -c[n]
        e[n] = -e[n]


def tridpss(a, b, c, d, e, f):
    """utils/linear3.f90:321-442 TRIDPSS. f: (nmax, len) in place."""
    nmax = a.shape[0]
    f[0] = f[0] * b[0]
    for n in range(1, nmax - 1):
        f[n] = f[n] * b[n] + a[n] * f[n - 1]
    wrk = np.zeros_like(f[0])
    for n in range(nmax - 1):
        wrk = wrk + d[n] * f[n]
    f[nmax - 1] = (f[nmax - 1] - wrk) * b[nmax - 1]
    f[nmax - 2] = e[nmax - 2] * f[nmax - 1] + f[nmax - 2]
    for n in range(nmax - 3, -1, -1):
        f[n] = f[n] + c[n] * f[n + 1] + e[n] * f[nmax - 1]


# ######################################################################################
# utils/linear5.f90 -- pentadiagonal LU in reverse ordering (CompactJacobian6Penta) and its periodic (Sherman-Morrison-Woodbury) form
# ######################################################################################
def pentadfs2(a, b, c, d, e):
    """utils/linear5.f90:156-203 PENTADFS2.  In place; arrays (nmax,)."""
    nmax = a.shape[0]
    n = nmax - 1
    e[n] = 1.0
    d[n] = 1.0
    n = nmax - 2
    e[n] = 1.0
    d[n] = d[n] / c[n + 1]
    c[n] = c[n] - d[n] * b[n + 1]
    b[n] = b[n] - d[n] * a[n + 1]
    for n in range(nmax - 3, 1, -1):
        e[n] = e[n] / c[n + 2]
        d[n] = (d[n] - e[n] * b[n + 2]) / c[n + 1]
        c[n] = c[n] - d[n] * b[n + 1] - e[n] * a[n + 2]
        b[n] = b[n] - d[n] * a[n + 1]
    n = 1
    e[n] = e[n] / c[n + 2]
    d[n] = (d[n] - e[n] * b[n + 2]) / c[n + 1]
    c[n] = c[n] - d[n] * b[n + 1] - e[n] * a[n + 2]
    b[n] = b[n] - d[n] * a[n + 1]
    a[n] = 1.0
    n = 0
    e[n] = e[n] / c[n + 2]
    d[n] = (d[n] - e[n] * b[n + 2]) / c[n + 1]
    c[n] = c[n] - d[n] * b[n + 1] - e[n] * a[n + 2]
    b[n] = 1.0
    a[n] = 1.0


def pentadss2(a, b, c, d, e, f):
    """utils/linear5.f90:207-244 PENTADSS2.  f: (nmax, len) in place."""
    nmax = a.shape[0]
    n = nmax - 2
    f[n] = f[n] - f[n + 1] * d[n]
    for n in range(nmax - 3, -1, -1):
        f[n] = f[n] - f[n + 1] * d[n] - f[n + 2] * e[n]
    f[0] = f[0] / c[0]
    f[1] = (f[1] - f[0] * b[1]) / c[1]
    for n in range(2, nmax):
        f[n] = (f[n] - f[n - 1] * b[n] - f[n - 2] * a[n]) / c[n]


def pentadpfs(a, b, c, d, e, f, g):
    """utils/linear5.f90:273-347 PENTADPFS.  In place; f, g are outputs (the two Woodbury vectors)."""
    nmax = a.shape[0]
    a0, b0, en, dn = a[0], b[0], e[nmax - 1], d[nmax - 1]
    b[1] = b[1] - d[nmax - 1]
    c[0] = c[0] - e[nmax - 1]
    c[1] = c[1] - e[nmax - 1]
    c[nmax - 2] = c[nmax - 2] - a[0]
    c[nmax - 1] = c[nmax - 1] - a[0]
    d[nmax - 2] = d[nmax - 2] - b[0]
    a[0] = 0.0; a[1] = 0.0; b[0] = 0.0
    d[nmax - 1] = 0.0; e[nmax - 1] = 0.0; e[nmax - 2] = 0.0
    pentadfs2(a, b, c, d, e)
    a[0] = a0; b[0] = b0; e[nmax - 1] = en; d[nmax - 1] = dn
    f[:] = 0.0; f[0] = 1.0; f[nmax - 2] = 1.0
    g[:] = 0.0; g[1] = 1.0; g[nmax - 1] = 1.0
    ff, gg = f.reshape(nmax, 1), g.reshape(nmax, 1)
    pentadss2(a, b, c, d, e, ff)
    pentadss2(a, b, c, d, e, gg)
    m1 = e[nmax - 1] * f[0] + a[0] * f[nmax - 2] + b[0] * f[nmax - 1] + 1.0
    m2 = e[nmax - 1] * g[0] + a[0] * g[nmax - 2] + b[0] * g[nmax - 1]
    m3 = d[nmax - 1] * f[0] + e[nmax - 1] * f[1] + a[0] * f[nmax - 1]
    m4 = d[nmax - 1] * g[0] + e[nmax - 1] * g[1] + a[0] * g[nmax - 1] + 1.0
    if (m1 * m4 - m2 * m3) < 1e-8:                     # :340-344: TLab_Stop(DNS_ERROR_PENTADP)
        raise ValueError("FDM_CreatePlan. Pendad - matrix M not invertible.")


def pentadpss(a, b, c, d, e, f, g, frc):
    """utils/linear5.f90:352-411 PENTADPSS.  frc: (nmax, len) in place."""
    nmax = a.shape[0]
    pentadss2(a, b, c, d, e, frc)
    m1 = e[nmax - 1] * f[0] + a[0] * f[nmax - 2] + b[0] * f[nmax - 1] + 1.0
    m2 = e[nmax - 1] * g[0] + a[0] * g[nmax - 2] + b[0] * g[nmax - 1]
    m3 = d[nmax - 1] * f[0] + e[nmax - 1] * f[1] + a[0] * f[nmax - 1]
    m4 = d[nmax - 1] * g[0] + e[nmax - 1] * g[1] + a[0] * g[nmax - 1] + 1.0
    di = 1 / (m1 * m4 - m2 * m3)
    d11 = di * (m4 * e[nmax - 1] - m2 * d[nmax - 1])
    d12 = di * (m4 * b[0] - m2 * a[0])
    d13 = di * m4 * a[0]
    d14 = di * m2 * e[nmax - 1]
    d21 = di * (m1 * d[nmax - 1] - m3 * e[nmax - 1])
    d22 = di * (m1 * a[0] - m3 * b[0])
    d23 = di * m3 * a[0]
    d24 = di * m1 * e[nmax - 1]
    dummy1 = d11 * frc[0] + d12 * frc[nmax - 1] + d13 * frc[nmax - 2] - d14 * frc[1]
    dummy2 = d21 * frc[0] + d22 * frc[nmax - 1] - d23 * frc[nmax - 2] + d24 * frc[1]
    for n in range(2, nmax - 3):                       # main loop, rows 3 .. nmax-3 (1-based)
        frc[n] = frc[n] - dummy1 * f[n] - dummy2 * g[n]
    for n in (0, 1, nmax - 3, nmax - 2, nmax - 1):     # boundaries, from the still unmodified rows 1, 2, nmax-1, nmax
        frc[n] = frc[n] - dummy1 * f[n] - dummy2 * g[n]


# ######################################################################################
# fdm/fdm_com1_jacobian.f90, fdm/fdm_com2_jacobian.f90 -- coefficient generation
# ######################################################################################
def _cshift(v, s):
    """Fortran cshift(v, s): result(i) = v(i+s) circularly."""
    return np.roll(v, -s)


def create_system_1der(dx, ndl, ndr, coef_int, coef_bc1=None, coef_bc2=None, coef_bc3=None):
    """fdm/fdm_com1_jacobian.f90:195-291 Create_System_1der."""
    nx = dx.shape[0]
    lhs = np.zeros((nx, ndl))
    rhs = np.zeros((nx, ndr))
    idl = ndl // 2 + 1
    idr = ndr // 2 + 1
    lhs[:, idl - 1] = 1.0
    for ic in range(1, idl):
        lhs[:, idl - ic - 1] = coef_int[ic - 1]
        lhs[:, idl + ic - 1] = coef_int[ic - 1]
    rhs[:, idr - 1] = 0.0
    for ic in range(1, idr):
        rhs[:, idr - ic - 1] = -coef_int[ic + 1]
        rhs[:, idr + ic - 1] = coef_int[ic + 1]

    if coef_bc1 is not None:
        n = 0
        lhs[n, :] = 0.0
        lhs[n, idl - 1] = 1.0
        if idl > 1:
            icmax = min(idl - 1, 2)
            lhs[n, idl:idl + icmax] = coef_bc1[0:icmax]
        rhs[n, :] = 0.0
        icmax = min(idr, 4)
        rhs[n, idr - 1:idr - 1 + icmax] = coef_bc1[2:2 + icmax]
        rhs[n, 0] = coef_bc1[2 + icmax]
        lhs[nx - 1, :] = lhs[0, ::-1]
        rhs[nx - 1, :] = -rhs[0, ::-1]

    if coef_bc2 is not None:
        n = 1
        if ndl == 3:
            lhs[n, :] = [coef_bc2[0], 1.0, coef_bc2[1]]
        elif ndl == 5:
            lhs[n, :] = [0.0, coef_bc2[0], 1.0, coef_bc2[1], 0.0]
        rhs[n, :] = 0.0
        icmax = min(idr + 1, 4)
        rhs[n, idr - 2:idr - 2 + icmax] = coef_bc2[2:2 + icmax]
        lhs[nx - 2, :] = lhs[1, ::-1]
        rhs[nx - 2, :] = -rhs[1, ::-1]

    if coef_bc3 is not None:
        n = 2
        if ndl == 5:
            lhs[n, :] = [0.0, coef_bc3[0], 1.0, coef_bc3[1], 0.0]
        rhs[n, :] = 0.0
        icmax = min(idr + 2, 6)
        rhs[n, idr - 3:idr - 3 + icmax] = coef_bc3[2:2 + icmax]
        lhs[nx - 3, :] = lhs[2, ::-1]
        rhs[nx - 3, :] = -rhs[2, ::-1]

    # multiply by the Jacobian (:279-284)
    lhs[:, idl - 1] = lhs[:, idl - 1] * dx
    for ic in range(1, idl):
        lhs[:, idl - ic - 1] = lhs[:, idl - ic - 1] * _cshift(dx, -ic)
        lhs[:, idl + ic - 1] = lhs[:, idl + ic - 1] * _cshift(dx, +ic)
    # normalize s.t. 1. upper-diagonal is 1 (:286-288)
    lhs = lhs / coef_int[2]
    rhs = rhs / coef_int[2]
    return lhs, rhs


def fdm_c1n4_jacobian(dx, periodic):
    """fdm/fdm_com1_jacobian.f90:38-83."""
    coef = np.array([0.25, 0.0, 0.75, 0.0, 0.0])
    if periodic:
        lhs, rhs = create_system_1der(dx, 3, 3, coef)
    else:
        bc1 = np.array([2.0, 0.0, -2.5, 2.0, 0.5, 0.0])
        lhs, rhs = create_system_1der(dx, 3, 3, coef, bc1)
    return lhs, rhs, (3, 3), coef


def fdm_c1n6_jacobian(dx, periodic):
    """fdm/fdm_com1_jacobian.f90:86-132."""
    coef = np.array([1.0 / 3.0, 0.0, 7.0 / 9.0, 1.0 / 36.0, 0.0])
    if periodic:
        lhs, rhs = create_system_1der(dx, 3, 5, coef)
    else:
        # REFERENCE DEFECT reproduced deliberately (the twin of the one in the C2N6-Hyper closure): with 7 RHS diagonals Create_System_1der reads
        # coef_bc1(3 + icmax) = coef_bc1(7) of a 6-element array (fdm_com1_jacobian.f90:237, icmax = 4); in the flang-built reference the next
        # stack slot is coef_bc2(1) = 1/6 (declared right after, :146), which becomes the "extended stencil" entry rhs(1,1) of the wall rows.
        bc1 = np.array([2.0, 0.0, -2.5, 2.0, 0.5, 0.0, PENTA_BC1_EXT])
        bc2 = np.array([1.0 / 6.0, 0.5, -5.0 / 9.0, -0.5, 1.0, 1.0 / 18.0])
        lhs, rhs = create_system_1der(dx, 3, 5, coef, bc1, bc2)
    return lhs, rhs, (3, 5), coef


def fdm_c1n6_jacobian_penta(dx, periodic):
    """fdm/fdm_com1_jacobian.f90:136-192 (FDM_C1N6_Jacobian_Penta): pentadiagonal LHS, 7-diagonal antisymmetric RHS, alpha = 0.56."""
    coef = np.zeros(5)
    coef[0] = 0.56
    coef[1] = 0.4 * (-1.0 / 3.0 + coef[0])
    coef[2] = 0.5 * (1.0 / 6.0) * (9.0 + coef[0] - 20.0 * coef[1])
    coef[3] = 0.25 * (1.0 / 15.0) * (-9.0 + 32.0 * coef[0] + 62.0 * coef[1])
    coef[4] = (1.0 / 6.0) * (1.0 / 10.0) * (1.0 - 3.0 * coef[0] + 12.0 * coef[1])
    if periodic:
        lhs, rhs = create_system_1der(dx, 5, 7, coef)
    else:
        # REFERENCE DEFECT reproduced deliberately (the twin of the one in the C2N6-Hyper closure): with 7 RHS diagonals Create_System_1der reads
        # coef_bc1(3 + icmax) = coef_bc1(7) of a 6-element array (fdm_com1_jacobian.f90:237, icmax = 4); in the flang-built reference the next
        # stack slot is coef_bc2(1) = 1/6 (declared right after, :146), which becomes the "extended stencil" entry rhs(1,1) of the wall rows.
        bc1 = np.array([2.0, 0.0, -2.5, 2.0, 0.5, 0.0, PENTA_BC1_EXT])
        bc2 = np.array([1.0 / 6.0, 0.5, -5.0 / 9.0, -0.5, 1.0, 1.0 / 18.0])
        bc3 = np.array([1.0 / 3.0, 1.0 / 3.0, -1.0 / 36.0, -7.0 / 9.0, 0.0, 7.0 / 9.0, 1.0 / 36.0, 0.0])
        lhs, rhs = create_system_1der(dx, 5, 7, coef, bc1, bc2, bc3)
    return lhs, rhs, (5, 7), coef


def create_system_2der(dx2, ndl, ndr, coef_int, coef_bc1=None, coef_bc2=None, coef_bc3=None):
    """fdm/fdm_com2_jacobian.f90:179-282 Create_System_2der. dx2: (nx, 2) = [dx/ds, d2x/ds2]."""
    nx = dx2.shape[0]
    lhs = np.zeros((nx, ndl))
    rhs = np.zeros((nx, ndr))
    rhs_d1 = np.zeros((nx, ndl))
    idl = ndl // 2 + 1
    idr = ndr // 2 + 1
    lhs[:, idl - 1] = 1.0
    for ic in range(1, idl):
        lhs[:, idl - ic - 1] = coef_int[ic - 1]
        lhs[:, idl + ic - 1] = coef_int[ic - 1]
    rhs[:, idr - 1] = 0.0
    for ic in range(1, idr):
        rhs[:, idr - 1] = rhs[:, idr - 1] - 2.0 * coef_int[ic + 1]
        rhs[:, idr - ic - 1] = coef_int[ic + 1]
        rhs[:, idr + ic - 1] = coef_int[ic + 1]

    if coef_bc1 is not None:
        n = 0
        lhs[n, :] = 0.0
        lhs[n, idl - 1] = 1.0
        if idl > 1:
            icmax = min(idl - 1, 2)
            lhs[n, idl:idl + icmax] = coef_bc1[0:icmax]
        rhs[n, :] = 0.0
        icmax = min(idr, 4)
        rhs[n, idr - 1:idr - 1 + icmax] = coef_bc1[2:2 + icmax]
        rhs[n, 0] = coef_bc1[2 + icmax]
        lhs[nx - 1, :] = lhs[0, ::-1]
        rhs[nx - 1, :] = rhs[0, ::-1]

    if coef_bc2 is not None:
        n = 1
        if ndl == 3:
            lhs[n, :] = [coef_bc2[0], 1.0, coef_bc2[1]]
        rhs[n, :] = 0.0
        icmax = min(idr + 1, 4)
        rhs[n, idr - 2:idr - 2 + icmax] = coef_bc2[2:2 + icmax]
        lhs[nx - 2, :] = lhs[1, ::-1]
        rhs[nx - 2, :] = rhs[1, ::-1]

    if coef_bc3 is not None:
        n = 2
        if ndl == 3:
            lhs[n, :] = [coef_bc3[0], 1.0, coef_bc3[1]]
        rhs[n, :] = 0.0
        icmax = min(idr + 2, 6)
        rhs[n, idr - 3:idr - 3 + icmax] = coef_bc3[2:2 + icmax]
        lhs[nx - 3, :] = lhs[2, ::-1]
        rhs[nx - 3, :] = rhs[2, ::-1]

    dx1 = dx2[:, 0]
    dxx = dx2[:, 1]
    # multiply by the Jacobians (:263-274)
    rhs_d1[:, idl - 1] = -lhs[:, idl - 1] * dxx
    for ic in range(1, idl):
        rhs_d1[:, idl - ic - 1] = -lhs[:, idl - ic - 1] * _cshift(dxx, -ic)
        rhs_d1[:, idl + ic - 1] = -lhs[:, idl + ic - 1] * _cshift(dxx, +ic)
    lhs[:, idl - 1] = lhs[:, idl - 1] * dx1 * dx1
    for ic in range(1, idl):
        lhs[:, idl - ic - 1] = lhs[:, idl - ic - 1] * _cshift(dx1, -ic) * _cshift(dx1, -ic)
        lhs[:, idl + ic - 1] = lhs[:, idl + ic - 1] * _cshift(dx1, +ic) * _cshift(dx1, +ic)
    lhs = lhs / coef_int[2]
    rhs = rhs / coef_int[2]
    rhs_d1 = rhs_d1 / coef_int[2]
    return lhs, rhs, rhs_d1


def fdm_c2n4_jacobian(dx2, periodic):
    """fdm/fdm_com2_jacobian.f90:40-84 (FDM_C2N4_Jacobian)."""
    coef = np.array([0.1, 0.0, 1.2, 0.0, 0.0])
    if periodic:
        lhs, rhs, rd1 = create_system_2der(dx2, 3, 5, coef)
    else:
        bc1 = np.array([11.0, 0.0, 13.0, -27.0, 15.0, -1.0])
        lhs, rhs, rd1 = create_system_2der(dx2, 3, 5, coef, bc1)
    return lhs, rhs, rd1, (3, 5), coef


def fdm_c2n6_jacobian(dx2, periodic):
    """fdm/fdm_com2_jacobian.f90:86-122 (FDM_C2N6_Jacobian)."""
    coef = np.array([2.0 / 11.0, 0.0, 12.0 / 11.0, 3.0 / 44.0, 0.0])
    if periodic:
        lhs, rhs, rd1 = create_system_2der(dx2, 3, 5, coef)
    else:
        bc1 = np.array([11.0, 0.0, 13.0, -27.0, 15.0, -1.0])
        bc2 = np.array([0.1, 0.1, 1.2, -2.4, 1.2, 0.0])
        lhs, rhs, rd1 = create_system_2der(dx2, 3, 5, coef, bc1, bc2)
    return lhs, rhs, rd1, (3, 5), coef


def fdm_c2n6_hyper_jacobian(dx2, periodic):
    """fdm/fdm_com2_jacobian.f90:125-176 (FDM_C2N6_Hyper_Jacobian)."""
    kc = PI ** 2.0
    coef = np.array([(272.0 - 45.0 * kc) / (416.0 - 90.0 * kc),
                     0.0,
                     (48.0 - 135.0 * kc) / (1664.0 - 360.0 * kc),
                     (528.0 - 81.0 * kc) / (208.0 - 45.0 * kc) / 4.0,
                     -(432.0 - 63.0 * kc) / (1664.0 - 360.0 * kc) / 9.0])
    if periodic:
        lhs, rhs, rd1 = create_system_2der(dx2, 3, 7, coef)
    else:
        bc1 = np.array([11.0, 0.0, 13.0, -27.0, 15.0, -1.0])
        bc2 = np.array([0.1, 0.1, 1.2, -2.4, 1.2, 0.0])
        bc3 = np.array([2.0 / 11.0, 2.0 / 11.0, 3.0 / 44.0, 12.0 / 11.0, -51.0 / 22.0, 12.0 / 11.0, 3.0 / 44.0, 0.0])
        # REFERENCE DEFECT reproduced deliberately: with 7 RHS diagonals Create_System_2der reads
        # coef_bc1(3 + icmax) = coef_bc1(7) of a 6-element array (fdm_com2_jacobian.f90:224, icmax = 4).
        # In the flang-built reference the next stack slot is coef_bc2(1) = 0.1 (declared right after,
        # fdm_com2_jacobian.f90:135), which becomes the "extended stencil" entry rhs(1,1) = -rhs... of the wall rows.
        # HYPER_BC1_EXT is that value; set it to 0.0 for the mathematically consistent closure.
        bc1 = np.append(bc1, HYPER_BC1_EXT)
        lhs, rhs, rd1 = create_system_2der(dx2, 3, 7, coef, bc1, bc2, bc3)
    return lhs, rhs, rd1, (3, 7), coef


# ######################################################################################
# fdm/fdm_base.f90 -- Neumann reduction
# ######################################################################################
def fdm_bcs_neumann(ibc, lhs, rhs, rhs_b, rhs_t):
    """fdm/fdm_base.f90:194-300 FDM_Bcs_Neumann. lhs modified in place; rhs_b (4,8), rhs_t (5,7) filled."""
    ndl = lhs.shape[1]
    idl = ndl // 2 + 1
    ndr = rhs.shape[1]
    idr = ndr // 2 + 1
    nx = lhs.shape[0]

    if ibc in (BCS_ND, BCS_NN):
        rhs_b[0:idr, 1:ndr + 1] = rhs[0:idr, 0:ndr]
        dummy = 1.0 / rhs[0, idr - 1]
        rhs_b[0, 1:ndr + 1] = -rhs_b[0, 1:ndr + 1] * dummy
        for ir in range(1, idr):
            for ic in range(idr + 1, ndr + 1):
                rhs_b[ir, ic - ir] = rhs_b[ir, ic - ir] + rhs_b[ir, idr - ir] * rhs_b[0, ic]
            ic = ndr + 1
            rhs_b[ir, ic - ir] = rhs_b[ir, ic - ir] + rhs_b[ir, idr - ir] * rhs_b[0, 1]
        lhs[0, :] = lhs[0, :] * dummy
        for ir in range(1, idr):
            for ic in range(idl + 1, ndl + 1):
                lhs[ir, ic - ir - 1] = lhs[ir, ic - ir - 1] - rhs_b[ir, idr - ir] * lhs[0, ic - 1]
            rhs_b[ir, idr - ir] = rhs_b[ir, idr - ir] * lhs[0, idl - 1]
        for ir in range(1, idl):
            rhs_b[ir, idr - ir] = rhs_b[ir, idr - ir] - lhs[ir, idl - ir - 1]
        rhs_b[0, idr] = lhs[0, idl - 1]

    if ibc in (BCS_DN, BCS_NN):
        rhs_t[1:idr + 1, 0:ndr] = rhs[nx - idr:nx, 0:ndr]
        dummy = 1.0 / rhs[nx - 1, idr - 1]
        rhs_t[idr, 0:ndr] = -rhs_t[idr, 0:ndr] * dummy
        for ir in range(1, idr):
            for ic in range(1, idr):
                rhs_t[idr - ir, ic + ir - 1] = rhs[nx - ir - 1, ic + ir - 1] + rhs[nx - ir - 1, idr + ir - 1] * rhs_t[idr, ic - 1]
            rhs_t[idr - ir, ir - 1] = rhs_t[idr - ir, ir - 1] + rhs[nx - ir - 1, idr + ir - 1] * rhs_t[idr, ndr - 1]
        lhs[nx - 1, :] = lhs[nx - 1, :] * dummy
        for ir in range(1, idr):
            for ic in range(1, idl):
                lhs[nx - ir - 1, ic + ir - 1] = lhs[nx - ir - 1, ic + ir - 1] - rhs[nx - ir - 1, idr + ir - 1] * lhs[nx - 1, ic - 1]
            rhs_t[idr - ir, idr + ir - 1] = rhs_t[idr - ir, idr + ir - 1] * lhs[nx - 1, idl - 1]
        for ir in range(1, idl):
            rhs_t[idr - ir, idr + ir - 1] = rhs_t[idr - ir, idr + ir - 1] - lhs[nx - ir - 1, idl + ir - 1]
        rhs_t[idr, idr - 1] = lhs[nx - 1, idl - 1]


# ######################################################################################
# fdm/fdm_matmul.f90 -- banded RHS products.  u, f: (n, nlines)
# ######################################################################################
def matmul_3d_add(rhs, u, f):
    """fdm/fdm_matmul.f90:126-153 MatMul_3d_add: f += B u, B tridiagonal with extended boundary stencil."""
    nx = rhs.shape[0]
    r1, r2, r3 = rhs[:, 0], rhs[:, 1], rhs[:, 2]
    f[0] = f[0] + u[0] * r2[0] + u[1] * r3[0] + u[2] * r1[0]
    for n in range(1, nx - 1):
        f[n] = f[n] + u[n - 1] * r1[n] + u[n] * r2[n] + u[n + 1] * r3[n]
    n = nx - 1
    f[n] = f[n] + u[n - 2] * r3[n] + u[n - 1] * r1[n] + u[n] * r2[n]


def matmul_3d(rhs, u, f, ibc=BCS_NONE, rhs_b=None, rhs_t=None):
    """fdm/fdm_matmul.f90:70-121 MatMul_3d. Returns (bcs_b, bcs_t) (None when not applicable)."""
    nx = rhs.shape[0]
    r1, r2, r3 = rhs[:, 0], rhs[:, 1], rhs[:, 2]
    bcs_b = bcs_t = None
    if ibc in (BCS_MIN, BCS_BOTH):
        bcs_b = f[0] * rhs_b[0, 2] + u[1] * rhs_b[0, 3] + u[2] * rhs_b[0, 1]
        f[1] = f[0] * rhs_b[1, 1] + u[1] * rhs_b[1, 2] + u[2] * rhs_b[1, 3]
        f[2] = f[0] * rhs_b[2, 0] + u[1] * rhs_b[2, 1] + u[2] * rhs_b[2, 2] + u[3] * rhs_b[2, 3]
    else:
        f[0] = u[0] * r2[0] + u[1] * r3[0] + u[2] * r1[0]
        f[1] = u[0] * r1[1] + u[1] * r2[1] + u[2] * r3[1]
        f[2] = u[1] * r1[2] + u[2] * r2[2] + u[3] * r3[2]
    for n in range(3, nx - 3):
        f[n] = u[n - 1] * r1[n] + u[n] * r2[n] + u[n + 1]
    if ibc in (BCS_MAX, BCS_BOTH):
        f[nx - 3] = u[nx - 4] * rhs_t[0, 0] + u[nx - 3] * rhs_t[0, 1] + u[nx - 2] * rhs_t[0, 2] + f[nx - 1] * rhs_t[0, 3]
        f[nx - 2] = u[nx - 3] * rhs_t[1, 0] + u[nx - 2] * rhs_t[1, 1] + f[nx - 1] * rhs_t[1, 2]
        bcs_t = u[nx - 3] * rhs_t[2, 2] + u[nx - 2] * rhs_t[2, 0] + f[nx - 1] * rhs_t[2, 1]
    else:
        f[nx - 3] = u[nx - 4] * r1[nx - 3] + u[nx - 3] * r2[nx - 3] + u[nx - 2] * r3[nx - 3]
        f[nx - 2] = u[nx - 3] * r1[nx - 2] + u[nx - 2] * r2[nx - 2] + u[nx - 1] * r3[nx - 2]
        f[nx - 1] = u[nx - 3] * r3[nx - 1] + u[nx - 2] * r1[nx - 1] + u[nx - 1] * r2[nx - 1]
    return bcs_b, bcs_t


def matmul_3d_antisym(rhs, u, f, ibc, rhs_b=None, rhs_t=None):
    """fdm/fdm_matmul.f90:157-212 MatMul_3d_antisym."""
    nx = rhs.shape[0]
    r1, r2, r3 = rhs[:, 0], rhs[:, 1], rhs[:, 2]
    if ibc == BCS_PERIODIC:
        f[0] = u[1] - u[nx - 1]
        f[1] = u[2] - u[0]
    elif ibc in (BCS_ND, BCS_NN):
        f[1] = f[0] * rhs_b[1, 1] + u[1] * rhs_b[1, 2] + u[2] * rhs_b[1, 3]
    else:
        f[0] = u[0] * r2[0] + u[1] * r3[0] + u[2] * r1[0]
        f[1] = u[0] * r1[1] + u[1] * r2[1] + u[2] * r3[1]
    for n in range(2, nx - 2):
        f[n] = u[n + 1] - u[n - 1]
    if ibc == BCS_PERIODIC:
        f[nx - 2] = u[nx - 1] - u[nx - 3]
        f[nx - 1] = u[0] - u[nx - 2]
    elif ibc in (BCS_DN, BCS_NN):
        f[nx - 2] = u[nx - 3] * rhs_t[1, 0] + u[nx - 2] * rhs_t[1, 1] + f[nx - 1] * rhs_t[1, 2]
    else:
        f[nx - 2] = u[nx - 3] * r1[nx - 2] + u[nx - 2] * r2[nx - 2] + u[nx - 1] * r3[nx - 2]
        f[nx - 1] = u[nx - 3] * r3[nx - 1] + u[nx - 2] * r1[nx - 1] + u[nx - 1] * r2[nx - 1]


def matmul_5d_antisym(rhs, u, f, ibc, rhs_b=None, rhs_t=None):
    """fdm/fdm_matmul.f90:359-419 MatMul_5d_antisym (default first-derivative RHS, C1N6)."""
    nx = rhs.shape[0]
    r1, r2, r3, r4, r5 = (rhs[:, k] for k in range(5))
    r5_loc = r5[3]
    if ibc == BCS_PERIODIC:
        f[0] = u[1] - u[nx - 1] + r5_loc * (u[2] - u[nx - 2])
        f[1] = u[2] - u[0] + r5_loc * (u[3] - u[nx - 1])
        f[2] = u[3] - u[1] + r5_loc * (u[4] - u[0])
    elif ibc in (BCS_ND, BCS_NN):
        # f[0] holds the boundary value (zeroed by the caller for homogeneous Neumann)
        f[1] = f[0] * rhs_b[1, 2] + u[1] * rhs_b[1, 3] + u[2] * rhs_b[1, 4] + u[3] * rhs_b[1, 5]
        f[2] = f[0] * rhs_b[2, 1] + u[1] * rhs_b[2, 2] + u[2] * rhs_b[2, 3] + u[3] * rhs_b[2, 4] + u[4] * rhs_b[2, 5]
    else:
        f[0] = u[0] * r3[0] + u[1] * r4[0] + u[2] * r5[0] + u[3] * r1[0]
        f[1] = u[0] * r2[1] + u[1] * r3[1] + u[2] * r4[1] + u[3] * r5[1]
        f[2] = u[0] * r1[2] + u[1] * r2[2] + u[2] * r3[2] + u[3] * r4[2] + u[4] * r5[2]
    for n in range(3, nx - 3):
        f[n] = u[n + 1] - u[n - 1] + r5_loc * (u[n + 2] - u[n - 2])
    if ibc == BCS_PERIODIC:
        f[nx - 3] = u[nx - 2] - u[nx - 4] + r5_loc * (u[nx - 1] - u[nx - 5])
        f[nx - 2] = u[nx - 1] - u[nx - 3] + r5_loc * (u[0] - u[nx - 4])
        f[nx - 1] = u[0] - u[nx - 2] + r5_loc * (u[1] - u[nx - 3])
    elif ibc in (BCS_DN, BCS_NN):
        f[nx - 3] = u[nx - 5] * rhs_t[1, 0] + u[nx - 4] * rhs_t[1, 1] + u[nx - 3] * rhs_t[1, 2] + u[nx - 2] * rhs_t[1, 3] + f[nx - 1] * rhs_t[1, 4]
        f[nx - 2] = u[nx - 4] * rhs_t[2, 0] + u[nx - 3] * rhs_t[2, 1] + u[nx - 2] * rhs_t[2, 2] + f[nx - 1] * rhs_t[2, 3]
    else:
        f[nx - 3] = u[nx - 5] * r1[nx - 3] + u[nx - 4] * r2[nx - 3] + u[nx - 3] * r3[nx - 3] + u[nx - 2] * r4[nx - 3] + u[nx - 1] * r5[nx - 3]
        f[nx - 2] = u[nx - 4] * r1[nx - 2] + u[nx - 3] * r2[nx - 2] + u[nx - 2] * r3[nx - 2] + u[nx - 1] * r4[nx - 2]
        f[nx - 1] = u[nx - 4] * r5[nx - 1] + u[nx - 3] * r1[nx - 1] + u[nx - 2] * r2[nx - 1] + u[nx - 1] * r3[nx - 1]


def matmul_7d_antisym(rhs, u, f, ibc, rhs_b=None, rhs_t=None):
    """fdm/fdm_matmul.f90:491-558 MatMul_7d_antisym (first-derivative RHS of CompactJacobian6Penta)."""
    nx = rhs.shape[0]
    r1, r2, r3, r4, r5, r6, r7 = (rhs[:, k] for k in range(7))
    r6_loc, r7_loc = r6[4], r7[4]
    if ibc == BCS_PERIODIC:
        f[0] = u[1] - u[nx - 1] + r6_loc * (u[2] - u[nx - 2]) + r7_loc * (u[3] - u[nx - 3])
        f[1] = u[2] - u[0] + r6_loc * (u[3] - u[nx - 1]) + r7_loc * (u[4] - u[nx - 2])
        f[2] = u[3] - u[1] + r6_loc * (u[4] - u[0]) + r7_loc * (u[5] - u[nx - 1])
        f[3] = u[4] - u[2] + r6_loc * (u[5] - u[1]) + r7_loc * (u[6] - u[0])
    elif ibc in (BCS_ND, BCS_NN):
        rb = rhs_b      # rhs_b(j, c) -> rb[j-1, c]
        f[1] = f[0] * rb[1, 3] + u[1] * rb[1, 4] + u[2] * rb[1, 5] + u[3] * rb[1, 6] + u[4] * rb[1, 7]
        f[2] = f[0] * rb[2, 2] + u[1] * rb[2, 3] + u[2] * rb[2, 4] + u[3] * rb[2, 5] + u[4] * rb[2, 6] + u[5] * rb[2, 7]
        f[3] = f[0] * rb[3, 1] + u[1] * rb[3, 2] + u[2] * rb[3, 3] + u[3] * rb[3, 4] + u[4] * rb[3, 5] + u[5] * rb[3, 6] + u[6] * rb[3, 7]
    else:
        f[0] = u[0] * r4[0] + u[1] * r5[0] + u[2] * r6[0] + u[3] * r7[0] + u[4] * r1[0]
        f[1] = u[0] * r3[1] + u[1] * r4[1] + u[2] * r5[1] + u[3] * r6[1] + u[4] * r7[1]
        f[2] = u[0] * r2[2] + u[1] * r3[2] + u[2] * r4[2] + u[3] * r5[2] + u[4] * r6[2] + u[5] * r7[2]
        f[3] = u[0] * r1[3] + u[1] * r2[3] + u[2] * r3[3] + u[3] * r4[3] + u[4] * r5[3] + u[5] * r6[3] + u[6] * r7[3]
    for n in range(4, nx - 4):
        f[n] = u[n + 1] - u[n - 1] + r6_loc * (u[n + 2] - u[n - 2]) + r7_loc * (u[n + 3] - u[n - 3])
    if ibc == BCS_PERIODIC:
        f[nx - 4] = u[nx - 3] - u[nx - 5] + r6_loc * (u[nx - 2] - u[nx - 6]) + r7_loc * (u[nx - 1] - u[nx - 7])
        f[nx - 3] = u[nx - 2] - u[nx - 4] + r6_loc * (u[nx - 1] - u[nx - 5]) + r7_loc * (u[0] - u[nx - 6])
        f[nx - 2] = u[nx - 1] - u[nx - 3] + r6_loc * (u[0] - u[nx - 4]) + r7_loc * (u[1] - u[nx - 5])
        f[nx - 1] = u[0] - u[nx - 2] + r6_loc * (u[1] - u[nx - 3]) + r7_loc * (u[2] - u[nx - 4])
    elif ibc in (BCS_DN, BCS_NN):
        rt = rhs_t      # rhs_t(r, c) -> rt[r, c-1]
        f[nx - 4] = u[nx - 7] * rt[1, 0] + u[nx - 6] * rt[1, 1] + u[nx - 5] * rt[1, 2] + u[nx - 4] * rt[1, 3] + u[nx - 3] * rt[1, 4] + u[nx - 2] * rt[1, 5] + f[nx - 1] * rt[1, 6]
        f[nx - 3] = u[nx - 6] * rt[2, 0] + u[nx - 5] * rt[2, 1] + u[nx - 4] * rt[2, 2] + u[nx - 3] * rt[2, 3] + u[nx - 2] * rt[2, 4] + f[nx - 1] * rt[2, 5]
        f[nx - 2] = u[nx - 5] * rt[3, 0] + u[nx - 4] * rt[3, 1] + u[nx - 3] * rt[3, 2] + u[nx - 2] * rt[3, 3] + f[nx - 1] * rt[3, 4]
    else:
        n = nx - 4
        f[n] = u[n - 3] * r1[n] + u[n - 2] * r2[n] + u[n - 1] * r3[n] + u[n] * r4[n] + u[n + 1] * r5[n] + u[n + 2] * r6[n] + u[n + 3] * r7[n]
        n = nx - 3
        f[n] = u[n - 3] * r1[n] + u[n - 2] * r2[n] + u[n - 1] * r3[n] + u[n] * r4[n] + u[n + 1] * r5[n] + u[n + 2] * r6[n]
        n = nx - 2
        f[n] = u[n - 3] * r1[n] + u[n - 2] * r2[n] + u[n - 1] * r3[n] + u[n] * r4[n] + u[n + 1] * r5[n]
        n = nx - 1
        f[n] = u[n - 4] * r7[n] + u[n - 3] * r1[n] + u[n - 2] * r2[n] + u[n - 1] * r3[n] + u[n] * r4[n]


def matmul_5d_sym(rhs, u, f, ibc):
    """fdm/fdm_matmul.f90:423-485 MatMul_5d_sym (C2N4/C2N6 second-derivative RHS)."""
    nx = rhs.shape[0]
    r1, r2, r3, r4, r5 = (rhs[:, k] for k in range(5))
    r5_loc = r5[2]
    r3_loc = r3[2]
    if ibc == BCS_PERIODIC:
        f[0] = r3_loc * u[0] + u[1] + u[nx - 1] + r5_loc * (u[2] + u[nx - 2])
        f[1] = r3_loc * u[1] + u[2] + u[0] + r5_loc * (u[3] + u[nx - 1])
    else:
        f[0] = u[0] * r3[0] + u[1] * r4[0] + u[2] * r5[0] + u[3] * r1[0]
        f[1] = u[0] * r2[1] + u[1] * r3[1] + u[2] * r4[1] + u[3] * r5[1]
        if ibc in (BCS_ND, BCS_NN):
            f[0] = 0.0
    for n in range(2, nx - 2):
        f[n] = r3_loc * u[n] + u[n + 1] + u[n - 1] + r5_loc * (u[n + 2] + u[n - 2])
    if ibc == BCS_PERIODIC:
        f[nx - 2] = r3_loc * u[nx - 2] + u[nx - 1] + u[nx - 3] + r5_loc * (u[0] + u[nx - 4])
        f[nx - 1] = r3_loc * u[nx - 1] + u[0] + u[nx - 2] + r5_loc * (u[1] + u[nx - 3])
    else:
        f[nx - 2] = u[nx - 4] * r1[nx - 2] + u[nx - 3] * r2[nx - 2] + u[nx - 2] * r3[nx - 2] + u[nx - 1] * r4[nx - 2]
        f[nx - 1] = u[nx - 4] * r5[nx - 1] + u[nx - 3] * r1[nx - 1] + u[nx - 2] * r2[nx - 1] + u[nx - 1] * r3[nx - 1]
        if ibc in (BCS_DN, BCS_NN):
            f[nx - 1] = 0.0


def matmul_5d(rhs, u, f, ibc=BCS_DD, rhs_b=None, rhs_t=None):
    """fdm/fdm_matmul.f90:265-319 MatMul_5d (B pentadiagonal with per-row coefficients, first upper diagonal = 1 in the interior;
    the direct schemes of fdm_comx_direct.f90).  f[0] / f[nx-1] carry the boundary value where ibc says so (:283-288, :303-308).
    Returns (bcs_b, bcs_t) (None when not applicable)."""
    nx = rhs.shape[0]
    r1, r2, r3, r4, r5 = (rhs[:, k] for k in range(5))
    bcs_b = bcs_t = None
    if ibc in (BCS_MIN, BCS_BOTH):
        b = rhs_b
        bcs_b = f[0] * b[0, 3] + u[1] * b[0, 4] + u[2] * b[0, 5] + u[3] * b[0, 1]
        f[1] = f[0] * b[1, 2] + u[1] * b[1, 3] + u[2] * b[1, 4] + u[3] * b[1, 5]
        f[2] = f[0] * b[2, 1] + u[1] * b[2, 2] + u[2] * b[2, 3] + u[3] * b[2, 4] + u[4] * b[2, 5]
        f[3] = f[0] * b[3, 0] + u[1] * b[3, 1] + u[2] * b[3, 2] + u[3] * b[3, 3] + u[4] * b[3, 4] + u[5] * b[3, 5]
    else:
        f[0] = u[0] * r3[0] + u[1] * r4[0] + u[2] * r5[0] + u[3] * r1[0]
        f[1] = u[0] * r2[1] + u[1] * r3[1] + u[2] * r4[1] + u[3] * r5[1]
        f[2] = u[0] * r1[2] + u[1] * r2[2] + u[2] * r3[2] + u[3] * r4[2] + u[4] * r5[2]
        f[3] = u[1] * r1[3] + u[2] * r2[3] + u[3] * r3[3] + u[4] * r4[3] + u[5] * r5[3]
    for n in range(4, nx - 4):
        f[n] = u[n - 2] * r1[n] + u[n - 1] * r2[n] + u[n] * r3[n] + u[n + 1] + u[n + 2] * r5[n]
    if ibc in (BCS_MAX, BCS_BOTH):
        t = rhs_t                      # rhs_t(0:, 1:) -> [row, col - 1]
        f[nx - 4] = u[nx - 6] * t[0, 0] + u[nx - 5] * t[0, 1] + u[nx - 4] * t[0, 2] + u[nx - 3] * t[0, 3] + u[nx - 2] * t[0, 4] + f[nx - 1] * t[0, 5]
        f[nx - 3] = u[nx - 5] * t[1, 0] + u[nx - 4] * t[1, 1] + u[nx - 3] * t[1, 2] + u[nx - 2] * t[1, 3] + f[nx - 1] * t[1, 4]
        f[nx - 2] = u[nx - 4] * t[2, 0] + u[nx - 3] * t[2, 1] + u[nx - 2] * t[2, 2] + f[nx - 1] * t[2, 3]
        bcs_t = u[nx - 4] * t[3, 4] + u[nx - 3] * t[3, 0] + u[nx - 2] * t[3, 1] + f[nx - 1] * t[3, 2]
    else:
        n = nx - 4
        f[n] = u[n - 2] * r1[n] + u[n - 1] * r2[n] + u[n] * r3[n] + u[n + 1] * r4[n] + u[n + 2] * r5[n]
        n = nx - 3
        f[n] = u[n - 2] * r1[n] + u[n - 1] * r2[n] + u[n] * r3[n] + u[n + 1] * r4[n] + u[n + 2] * r5[n]
        n = nx - 2
        f[n] = u[n - 2] * r1[n] + u[n - 1] * r2[n] + u[n] * r3[n] + u[n + 1] * r4[n]
        n = nx - 1
        f[n] = u[n - 3] * r5[n] + u[n - 2] * r1[n] + u[n - 1] * r2[n] + u[n] * r3[n]
    return bcs_b, bcs_t


def matmul_7d_sym(rhs, u, f, ibc):
    """fdm/fdm_matmul.f90:562-642 MatMul_7d_sym (default second-derivative RHS, C2N6-Hyper)."""
    nx = rhs.shape[0]
    r1, r2, r3, r4, r5, r6, r7 = (rhs[:, k] for k in range(7))
    r7_loc = r7[3]
    r6_loc = r6[3]
    r4_loc = r4[3]
    if ibc == BCS_PERIODIC:
        f[0] = r4_loc * u[0] + u[1] + u[nx - 1] + r6_loc * (u[2] + u[nx - 2]) + r7_loc * (u[3] + u[nx - 3])
        f[1] = r4_loc * u[1] + u[2] + u[0] + r6_loc * (u[3] + u[nx - 1]) + r7_loc * (u[4] + u[nx - 2])
        f[2] = r4_loc * u[2] + u[3] + u[1] + r6_loc * (u[4] + u[0]) + r7_loc * (u[5] + u[nx - 1])
    else:
        f[0] = u[0] * r4[0] + u[1] * r5[0] + u[2] * r6[0] + u[3] * r7[0] + u[4] * r1[0]
        f[1] = u[0] * r3[1] + u[1] * r4[1] + u[2] * r5[1] + u[3] * r6[1] + u[4] * r7[1]
        f[2] = u[0] * r2[2] + u[1] * r3[2] + u[2] * r4[2] + u[3] * r5[2] + u[4] * r6[2] + u[5] * r7[2]
        if ibc in (BCS_ND, BCS_NN):
            f[0] = 0.0
    for n in range(3, nx - 3):
        f[n] = r4_loc * u[n] + u[n + 1] + u[n - 1] + r6_loc * (u[n + 2] + u[n - 2]) + r7_loc * (u[n + 3] + u[n - 3])
    if ibc == BCS_PERIODIC:
        f[nx - 3] = r4_loc * u[nx - 3] + u[nx - 2] + u[nx - 4] + r6_loc * (u[nx - 1] + u[nx - 5]) + r7_loc * (u[0] + u[nx - 6])
        f[nx - 2] = r4_loc * u[nx - 2] + u[nx - 1] + u[nx - 3] + r6_loc * (u[0] + u[nx - 4]) + r7_loc * (u[1] + u[nx - 5])
        f[nx - 1] = r4_loc * u[nx - 1] + u[0] + u[nx - 2] + r6_loc * (u[1] + u[nx - 3]) + r7_loc * (u[2] + u[nx - 4])
    else:
        f[nx - 3] = u[nx - 6] * r1[nx - 3] + u[nx - 5] * r2[nx - 3] + u[nx - 4] * r3[nx - 3] + u[nx - 3] * r4[nx - 3] + u[nx - 2] * r5[nx - 3] + u[nx - 1] * r6[nx - 3]
        f[nx - 2] = u[nx - 5] * r1[nx - 2] + u[nx - 4] * r2[nx - 2] + u[nx - 3] * r3[nx - 2] + u[nx - 2] * r4[nx - 2] + u[nx - 1] * r5[nx - 2]
        f[nx - 1] = u[nx - 5] * r7[nx - 1] + u[nx - 4] * r1[nx - 1] + u[nx - 3] * r2[nx - 1] + u[nx - 2] * r3[nx - 1] + u[nx - 1] * r4[nx - 1]
        if ibc in (BCS_DN, BCS_NN):
            f[nx - 1] = 0.0


# ######################################################################################
# fdm/fdm_derivative.f90 -- plans and batched 1-D solves
# ######################################################################################
class DerPlan:
    """fdm/fdm_derivative.f90:16-29 type fdm_derivative_dt."""

    def __init__(self, mode_fdm):
        self.mode_fdm = mode_fdm
        self.size = 0
        self.periodic = False
        self.need_1der = False
        self.nb_diag = (0, 0)
        self.rhs_b = np.zeros((4, 8))
        self.rhs_t = np.zeros((5, 7))
        self.lhs = self.rhs = self.mwn = self.lu = None


def _wavenumbers(nx):
    i = np.arange(1, nx + 1)
    return np.where(i <= nx // 2 + 1, 2.0 * PI * (i - 1) / nx, 2.0 * PI * (i - 1 - nx) / nx)


def der1_create_system(g, dx, periodic):
    """fdm/fdm_derivative.f90:146-214 FDM_Der1_CreateSystem."""
    nx = dx.shape[0]
    g.size = nx
    g.periodic = periodic
    if g.mode_fdm == FDM_COM4_JACOBIAN:
        lhs, rhs, nb, coef = fdm_c1n4_jacobian(dx, periodic)
    elif g.mode_fdm in (FDM_COM6_JACOBIAN, FDM_COM6_JACOBIAN_HYPER):
        lhs, rhs, nb, coef = fdm_c1n6_jacobian(dx, periodic)
    elif g.mode_fdm == FDM_COM6_JACOBIAN_PENTA:
        lhs, rhs, nb, coef = fdm_c1n6_jacobian_penta(dx, periodic)
    else:
        raise NotImplementedError("oracle: first-derivative scheme %d" % g.mode_fdm)
    g.lhs = np.zeros((nx, 5)); g.lhs[:, :nb[0]] = lhs
    g.rhs = np.zeros((nx, 7)); g.rhs[:, :nb[1]] = rhs
    g.nb_diag = nb
    g.mwn = np.zeros(nx)
    if periodic:
        wn = _wavenumbers(nx)
        # NOTE cos(wn) with coef(2) reproduces the reference (fdm_derivative.f90:207), see SURVEY 0.5
        g.mwn = 2.0 * (coef[2] * np.sin(wn) + coef[3] * np.sin(2.0 * wn) + coef[4] * np.sin(3.0 * wn)) \
            / (1.0 + 2.0 * coef[0] * np.cos(wn) + 2.0 * coef[1] * np.cos(wn))


def der1_initialize(g, dx, periodic, bcs_cases):
    """fdm/fdm_derivative.f90:63-142 FDM_Der1_Initialize."""
    der1_create_system(g, dx, periodic)
    nx, ndl, ndr = g.size, g.nb_diag[0], g.nb_diag[1]
    if periodic:
        g.lu = np.zeros((nx, ndl + 2))
        g.lu[:, :ndl] = g.lhs[:, :ndl]
        cols = [g.lu[:, k].copy() for k in range(ndl + 2)]
        if ndl == 3:
            tridpfs(*cols)
        else:
            pentadpfs(*cols)                       # fdm_derivative.f90:90-91
        for k in range(ndl + 2):
            g.lu[:, k] = cols[k]
    else:
        g.lu = np.zeros((nx, 20))
        for ib, bc in enumerate(bcs_cases):
            ip = ib * 5
            blk = g.lhs[:, :ndl].copy()
            fdm_bcs_neumann(bc, blk, g.rhs[:, :ndr], g.rhs_b, g.rhs_t)
            nmin, nmax = 0, nx
            if bc in (BCS_ND, BCS_NN):
                nmin += 1
            if bc in (BCS_DN, BCS_NN):
                nmax -= 1
            cols = [blk[nmin:nmax, k].copy() for k in range(ndl)]
            if ndl == 3:
                tridfs(*cols)
            else:
                pentadfs2(*cols)                   # :112-113
            for k in range(ndl):
                blk[nmin:nmax, k] = cols[k]
            g.lu[:, ip:ip + ndl] = blk


def der1_matmul(g, u, f, ibc):
    if getattr(g, "direct", False):          # FDM_COM4_DIRECT / FDM_COM6_DIRECT: g%matmul => MatMul_3d / MatMul_5d (fdm_derivative.f90:123-129)
        if g.nb_diag[1] == 3:
            matmul_3d(g.rhs, u, f, ibc, g.rhs_b, g.rhs_t)
        elif g.nb_diag[1] == 5:
            matmul_5d(g.rhs, u, f, ibc, g.rhs_b, g.rhs_t)
        else:
            raise NotImplementedError
    elif g.nb_diag[1] == 3:
        matmul_3d_antisym(g.rhs, u, f, ibc, g.rhs_b, g.rhs_t)
    elif g.nb_diag[1] == 5:
        matmul_5d_antisym(g.rhs, u, f, ibc, g.rhs_b, g.rhs_t)
    elif g.nb_diag[1] == 7:
        matmul_7d_antisym(g.rhs, u, f, ibc, g.rhs_b, g.rhs_t)
    else:
        raise NotImplementedError


def der1_solve(g, ibc, u, lu1=None):
    """fdm/fdm_derivative.f90:218-278 FDM_Der1_Solve. u: (n, nlines) -> result (n, nlines)."""
    lu1 = g.lu if lu1 is None else lu1
    n = g.size
    res = np.empty_like(u)
    ibc_loc = ibc
    ip = ibc_loc * 5
    if g.periodic:
        ibc_loc = BCS_PERIODIC
    nmin, nmax = 0, n
    if ibc_loc in (BCS_ND, BCS_NN):
        res[0] = 0.0
        nmin += 1
    if ibc_loc in (BCS_DN, BCS_NN):
        res[n - 1] = 0.0
        nmax -= 1
    der1_matmul(g, u, res, ibc_loc)
    if g.periodic:
        if g.nb_diag[0] == 3:
            tridpss(lu1[:, 0], lu1[:, 1], lu1[:, 2], lu1[:, 3], lu1[:, 4], res)
        else:
            pentadpss(*(lu1[:, k] for k in range(7)), res)
    else:
        if g.nb_diag[0] == 3:
            tridss(lu1[nmin:nmax, ip], lu1[nmin:nmax, ip + 1], lu1[nmin:nmax, ip + 2], res[nmin:nmax])
        else:
            pentadss2(*(lu1[nmin:nmax, ip + k] for k in range(5)), res[nmin:nmax])
    return res


def der2_create_system(g, dx2, periodic, uniform):
    """fdm/fdm_derivative.f90:337-409 FDM_Der2_CreateSystem."""
    nx = dx2.shape[0]
    g.size = nx
    g.periodic = periodic
    if g.mode_fdm == FDM_COM4_JACOBIAN:
        lhs, rhs, rd1, nb, coef = fdm_c2n4_jacobian(dx2, periodic)
        if not uniform:
            g.need_1der = True
    elif g.mode_fdm in (FDM_COM6_JACOBIAN, FDM_COM6_JACOBIAN_PENTA):
        lhs, rhs, rd1, nb, coef = fdm_c2n6_jacobian(dx2, periodic)
        if not uniform:
            g.need_1der = True
    elif g.mode_fdm == FDM_COM6_JACOBIAN_HYPER:
        lhs, rhs, rd1, nb, coef = fdm_c2n6_hyper_jacobian(dx2, periodic)
        if not uniform:
            g.need_1der = True
    else:
        raise NotImplementedError("oracle: second-derivative scheme %d" % g.mode_fdm)
    g.lhs = np.zeros((nx, 5)); g.lhs[:, :nb[0]] = lhs
    g.rhs = np.zeros((nx, 12)); g.rhs[:, :nb[1]] = rhs; g.rhs[:, nb[1]:nb[1] + 3] = rd1
    g.nb_diag = nb
    g.mwn = np.zeros(nx)
    if periodic:
        wn = _wavenumbers(nx)
        g.mwn = 2.0 * (coef[2] * (1.0 - np.cos(wn)) + coef[3] * (1.0 - np.cos(2.0 * wn)) + coef[4] * (1.0 - np.cos(3.0 * wn))) \
            / (1.0 + 2.0 * coef[0] * np.cos(wn) + 2.0 * coef[1] * np.cos(2.0 * wn))


def der2_initialize(g, dx2, periodic, uniform):
    """fdm/fdm_derivative.f90:282-333 FDM_Der2_Initialize."""
    der2_create_system(g, dx2, periodic, uniform)
    nx, ndl = g.size, g.nb_diag[0]
    if periodic:
        g.lu = np.zeros((nx, ndl + 2))
        g.lu[:, :ndl] = g.lhs[:, :ndl]
        cols = [g.lu[:, k].copy() for k in range(5)]
        tridpfs(*cols)
        for k in range(5):
            g.lu[:, k] = cols[k]
    else:
        g.lu = np.zeros((nx, ndl))
        g.lu[:, :ndl] = g.lhs[:, :ndl]
        cols = [g.lu[:, k].copy() for k in range(3)]
        tridfs(*cols)
        for k in range(3):
            g.lu[:, k] = cols[k]


def der2_solve(g, lu, u, du):
    """fdm/fdm_derivative.f90:413-459 FDM_Der2_Solve. lu is an argument (Burgers passes the nu-scaled one)."""
    res = np.empty_like(u)
    ibc = BCS_PERIODIC if g.periodic else BCS_DD
    if getattr(g, "direct", False):          # FDM_COM4_DIRECT / FDM_COM6_DIRECT: g%matmul => MatMul_5d (fdm_derivative.f90:318-323)
        matmul_5d(g.rhs, u, res, ibc)
    elif g.nb_diag[1] == 5:
        matmul_5d_sym(g.rhs, u, res, ibc)
    elif g.nb_diag[1] == 7:
        matmul_7d_sym(g.rhs, u, res, ibc)
    else:
        raise NotImplementedError
    if g.need_1der:
        ip = g.nb_diag[1]
        matmul_3d_add(g.rhs[:, ip:ip + 3], du, res)
    if g.periodic:
        tridpss(lu[:, 0], lu[:, 1], lu[:, 2], lu[:, 3], lu[:, 4], res)
    else:
        tridss(lu[:, 0], lu[:, 1], lu[:, 2], res)
    return res


# ######################################################################################
# fdm/fdm.f90 -- plan for one direction
# ######################################################################################
class FdmPlan:
    """fdm/fdm.f90:14-29 type fdm_dt."""

    def __init__(self, nodes, periodic, uniform, mode1=FDM_COM6_JACOBIAN, mode2=FDM_COM6_JACOBIAN_HYPER, hyper_bc1_ext=None, stagger=False):
        """fdm/fdm.f90:143-252 FDM_CreatePlan.  hyper_bc1_ext: None = the module's HYPER_BC1_EXT (what the flang-built reference reads).
        stagger: TLab_WorkFlow::stagger_on (horizontal pressure staggering): periodic directions get g%intl and the interpolatory der1%mwn (:236-248)."""
        self.stagger = bool(stagger)
        global HYPER_BC1_EXT
        saved = HYPER_BC1_EXT
        if hyper_bc1_ext is not None:
            HYPER_BC1_EXT = float(hyper_bc1_ext)
        try:
            self._create(nodes, periodic, uniform, mode1, mode2)
        finally:
            HYPER_BC1_EXT = saved

    def _create(self, nodes, periodic, uniform, mode1, mode2):
        nodes = np.asarray(nodes, dtype=np.float64)
        if periodic and mode1 == FDM_COM4_DIRECT: mode1 = FDM_COM4_JACOBIAN
        if periodic and mode1 == FDM_COM6_DIRECT: mode1 = FDM_COM6_JACOBIAN
        if periodic and mode2 == FDM_COM4_DIRECT: mode2 = FDM_COM4_JACOBIAN
        if periodic and mode2 == FDM_COM6_DIRECT: mode2 = FDM_COM6_JACOBIAN_HYPER
        nx = nodes.shape[0]
        self.size = nx
        self.periodic = periodic
        self.uniform = uniform
        self.der1 = DerPlan(mode1)
        self.der2 = DerPlan(mode2)
        self.jac = np.ones((nx, 3))
        if nx > 1:
            self.scale = nodes[nx - 1] - nodes[0]
            if periodic:
                self.scale = self.scale * (1.0 + 1.0 / float(nx - 1))
        else:
            self.scale = 1.0
            self.nodes = nodes.copy()
            return

        # first-order derivative: Jacobian from a unit-grid derivative of the node positions (:194-201)
        self.jac[:, 0] = 1.0
        der1_initialize(self.der1, self.jac[:, 0].copy(), False, [BCS_DD])
        self.der1.periodic = False
        self.jac[:, 0] = der1_solve(self.der1, BCS_NONE, nodes.reshape(nx, 1))[:, 0]
        self.nodes = nodes.copy()
        der1_initialize(self.der1, self.jac[:, 0].copy(), periodic, [BCS_DD, BCS_ND, BCS_DN, BCS_NN])
        if periodic:
            self.der1.mwn = self.der1.mwn / self.jac[0, 0]

        # second-order derivative (:212-233)
        self.jac[:, 1] = 1.0
        self.jac[:, 2] = 0.0
        der2_initialize(self.der2, self.jac[:, 1:3].copy(), False, True)
        self.der2.periodic = False
        self.jac[:, 2] = der2_solve(self.der2, self.der2.lu, nodes.reshape(nx, 1), self.jac[:, 1].reshape(nx, 1))[:, 0]
        self.jac[:, 1] = self.jac[:, 0]
        der2_initialize(self.der2, self.jac[:, 1:3].copy(), periodic, uniform)
        if self.der2.periodic:
            self.der2.mwn = self.der2.mwn / (self.jac[0, 0] ** 2)
        if getattr(self, "stagger", False) and periodic:                          # :236-248
            self.intl = Interpol()
            self.der1.mwn = interpol_initialize(nodes, self.jac[:, 0], self.intl)

    @classmethod
    def from_tables(cls, tab, periodic=False, mode1=FDM_COM6_JACOBIAN, mode2=FDM_COM6_DIRECT):
        """Plan from coefficient tables the reference built (tests/golden: the direct schemes of fdm_comx_direct.f90 are not restated;
        their tables come from FDM_CreatePlan itself).  tab: dict as oracle/ref_lib.fdm_arrays."""
        self = cls.__new__(cls)
        n = tab["lhs1"].shape[0]
        self.size, self.periodic, self.uniform = n, periodic, not bool(tab["need_1der"]) and mode2 not in (FDM_COM4_DIRECT, FDM_COM6_DIRECT)
        self.jac = np.array(tab["jac"], dtype=np.float64)
        self.nodes = np.array(tab["nodes"], dtype=np.float64) if "nodes" in tab else None
        self.der1, self.der2 = DerPlan(mode1), DerPlan(mode2)
        for d, k in ((self.der1, "1"), (self.der2, "2")):
            d.size, d.periodic = n, periodic
            d.nb_diag = (int(tab["ndl" + k]), int(tab["ndr" + k]))
            d.lhs = np.array(tab["lhs" + k], dtype=np.float64)
            d.rhs = np.array(tab["rhs" + k], dtype=np.float64)
            d.lu = np.array(tab["lu" + k], dtype=np.float64)
            d.mwn = np.array(tab["mwn" + k], dtype=np.float64)
        self.der1.rhs_b = np.array(tab["rhs_b1"], dtype=np.float64)
        self.der1.rhs_t = np.array(tab["rhs_t1"], dtype=np.float64)
        self.der2.need_1der = bool(tab["need_1der"])
        self.der2.direct = mode2 in (FDM_COM4_DIRECT, FDM_COM6_DIRECT)
        self.der1.direct = mode1 in (FDM_COM4_DIRECT, FDM_COM6_DIRECT)
        return self

    def diffusion_lu(self, nu):
        """physics/opr_burgers.f90:100-111: LU of the second derivative with the diffusivity folded in."""
        lu = self.der2.lu.copy()
        if self.periodic:
            lu[:, 1] = self.der2.lu[:, 1] * nu
            lu[:, 3] = self.der2.lu[:, 3] / nu
        else:
            lu[:, 1] = self.der2.lu[:, 1] * nu
            lu[:, 2] = self.der2.lu[:, 2] / nu
        return lu


# ######################################################################################
# operators/opr_partial.f90, physics/opr_burgers.f90 -- 3-D operators on flat x-fastest fields
# ######################################################################################
def _to_lines(u, nx, ny, nz, idir):
    """Arrange a flat field as (n, nlines) along idir exactly as the reference's local transposes do.
    X: utils/tlab_transpose.f90 via opr_partial.f90:87 -> (nyz, nx) lines-fastest == C shape (nx, ny*nz)
    Y: opr_partial.f90:303 -> b(nz, nxy) viewed (nx*nz, ny): line index = k + nz*i == C shape (ny, nx, nz)
    Z: none -> C shape (nz, nx*ny)."""
    a = u.reshape(nz, ny, nx)
    if idir == 1:
        return np.ascontiguousarray(a.transpose(2, 0, 1)).reshape(nx, nz * ny)   # [i][k][j] -> line = j + ny*k
    if idir == 2:
        return np.ascontiguousarray(a.transpose(1, 2, 0)).reshape(ny, nx * nz)   # [j][i][k] -> line = k + nz*i
    return a.reshape(nz, ny * nx)


def _from_lines(r, nx, ny, nz, idir):
    if idir == 1:
        return np.ascontiguousarray(r.reshape(nx, nz, ny).transpose(1, 2, 0)).ravel()
    if idir == 2:
        return np.ascontiguousarray(r.reshape(ny, nx, nz).transpose(2, 0, 1)).ravel()
    return np.ascontiguousarray(r).ravel()


# ######################################################################################
# fdm/fdm_interpolate.f90 + fdm/fdm_com0_jacobian.f90 -- interpolation between the velocity and the staggered pressure grid (periodic directions)
# ######################################################################################
class Interpol:
    """type(fdm_interpol_dt), fdm_interpolate.f90:14-21"""
    lu0i = lu1i = None


def fdm_c0int6p_lhs(n):
    """fdm_com0_jacobian.f90:29-44"""
    return np.full(n, 2.0 / 5.0), np.full(n, 4.0 / 3.0), np.full(n, 2.0 / 5.0)


def fdm_c1int6p_lhs(dx):
    """fdm_com0_jacobian.f90:287-320 (with the Jacobian multiplication)"""
    n = dx.shape[0]
    a, b, c = np.full(n, 9.0 / 63.0), np.full(n, 62.0 / 63.0), np.full(n, 9.0 / 63.0)
    c[n - 1] = c[n - 1] * dx[0]; b[0] = b[0] * dx[0]; a[1] = a[1] * dx[0]
    for i in range(1, n - 1):
        c[i - 1] = c[i - 1] * dx[i]; b[i] = b[i] * dx[i]; a[i + 1] = a[i + 1] * dx[i]
    c[n - 2] = c[n - 2] * dx[n - 1]; b[n - 1] = b[n - 1] * dx[n - 1]; a[0] = a[0] * dx[n - 1]
    return a, b, c


def _cyc(u, k):
    """u[(i + k) mod n] for every row i (the mod(...) index arithmetic of the periodic right-hand sides)"""
    return np.roll(u, -k, axis=0)


def fdm_c0intvp6p_rhs(u):
    """:50-76"""
    return _cyc(u, 1) + u + (1.0 / 15.0) * (_cyc(u, 2) + _cyc(u, -1))


def fdm_c0intpv6p_rhs(u):
    """:82-108"""
    return u + _cyc(u, -1) + (1.0 / 15.0) * (_cyc(u, 1) + _cyc(u, -2))


def fdm_c1intvp6p_rhs(u):
    """:326-353"""
    return (_cyc(u, 1) - u) + (17.0 / 189.0) * (_cyc(u, 2) - _cyc(u, -1))


def fdm_c1intpv6p_rhs(u):
    """:359-386"""
    return (u - _cyc(u, -1)) + (17.0 / 189.0) * (_cyc(u, 1) - _cyc(u, -2))


def interpol_initialize(x, dx, var):
    """fdm_interpolate.f90:33-96 FDM_Interpol_Initialize: LU of the two periodic tridiagonal systems; returns the interpolatory modified wavenumbers"""
    nx = x.shape[0]
    for name, (a, b, c) in (("lu0i", fdm_c0int6p_lhs(nx)), ("lu1i", fdm_c1int6p_lhs(np.asarray(dx, dtype=np.float64)))):
        lu = np.zeros((nx, 5))
        lu[:, 0], lu[:, 1], lu[:, 2] = a, b, c
        cols = [lu[:, k].copy() for k in range(5)]
        tridpfs(*cols)
        setattr(var, name, np.stack(cols, axis=1))
    wn = _wavenumbers(nx)
    c1, c3, c4 = 9.0 / 62.0, 63.0 / 62.0, 17.0 / 62.0
    wn = 2.0 * (c3 * np.sin(1.0 / 2.0 * wn) + c4 / 3.0 * np.sin(3.0 / 2.0 * wn)) / (1.0 + 2.0 * c1 * np.cos(wn))
    return wn / dx[0]


def fdm_interpol(direction, g, u, der):
    """FDM_Interpol (der = False) / FDM_Interpol_Der1 (der = True), fdm_interpolate.f90:98-160; direction 0: velocity -> pressure grid, 1: back"""
    if der:
        r = fdm_c1intvp6p_rhs(u) if direction == 0 else fdm_c1intpv6p_rhs(u)
        lu = g.lu1i
    else:
        r = fdm_c0intvp6p_rhs(u) if direction == 0 else fdm_c0intpv6p_rhs(u)
        lu = g.lu0i
    r = np.ascontiguousarray(r)
    tridpss(lu[:, 0], lu[:, 1], lu[:, 2], lu[:, 3], lu[:, 4], r)
    return r


def opr_partial(idir, itype, nx, ny, nz, ibc, g, u):
    """operators/opr_partial.f90:31-150 (X), :266-377 (Y), :154-262 (Z), serial branch.
    Returns (result, tmp1) with tmp1 = first derivative for OPR_P2_P1 (else None)."""
    n = (nx, ny, nz)[idir - 1]
    if n == 1 and idir != 1:
        z = np.zeros(nx * ny * nz)
        return z, (z.copy() if itype == OPR_P2_P1 else None)
    ul = _to_lines(np.asarray(u, dtype=np.float64), nx, ny, nz, idir)
    if itype == OPR_P1:
        return _from_lines(der1_solve(g.der1, ibc, ul), nx, ny, nz, idir), None
    if itype == OPR_P2:
        du = der1_solve(g.der1, ibc, ul) if g.der2.need_1der else np.zeros_like(ul)
        return _from_lines(der2_solve(g.der2, g.der2.lu, ul, du), nx, ny, nz, idir), None
    if itype == OPR_P2_P1:
        du = der1_solve(g.der1, ibc, ul)
        r = der2_solve(g.der2, g.der2.lu, ul, du)
        return _from_lines(r, nx, ny, nz, idir), _from_lines(du, nx, ny, nz, idir)
    if itype in (OPR_P0_INT_VP, OPR_P0_INT_PV, OPR_P1_INT_VP, OPR_P1_INT_PV) and idir in (1, 3):      # opr_partial.f90:110-120, :228-238
        r = fdm_interpol(0 if itype in (OPR_P0_INT_VP, OPR_P1_INT_VP) else 1, g.intl, ul, itype in (OPR_P1_INT_VP, OPR_P1_INT_PV))
        return _from_lines(r, nx, ny, nz, idir), None
    raise NotImplementedError


def opr_burgers(idir, nx, ny, nz, ibc, g, nu, s, vel, anelastic=None, dealiasing=None):
    """physics/opr_burgers.f90:190-273 (X), :277-355 (Y), :359-431 (Z) + OPR_Burgers_1D :439-521
    (serial, no dealiasing): result = nu d2s - vel ds along idir.
    anelastic = (rbackground, ribackground) (ny values each; nse_eqns == DNS_EQNS_ANELASTIC, opr_burgers.f90:128-183): along x and z the
    diffusion term is multiplied by rhoinv%values(line) = ribackground(y index of the line) (:134-151, :164-181, :504-507); along y the
    correction sits in the LU factors of the diffusion system (:153-160): U's inverse diagonal times ribackground, its superdiagonal times
    rbackground(2:).
    dealiasing = a tlab_oracle_filter.Filter of this direction ([Dealiasing], :478-500): the velocity and ds/dx are filtered along the line
    (OPR_FILTER_1D) before their product.
    Returns (result, s_transposed) with s_transposed the flat (lines-fastest) operand the reference leaves in tmp1."""
    n = (nx, ny, nz)[idir - 1]
    if n == 1:
        return np.zeros(nx * ny * nz), None
    sl = _to_lines(np.asarray(s, dtype=np.float64), nx, ny, nz, idir)
    vl = _to_lines(np.asarray(vel, dtype=np.float64), nx, ny, nz, idir)
    dsdx = der1_solve(g.der1, ibc, sl)
    lu = g.diffusion_lu(nu)
    if anelastic is not None and idir == 2:
        rb, ri = (np.asarray(a, dtype=np.float64) for a in anelastic)
        lu[:, 1] = lu[:, 1] * ri
        lu[:n - 1, 2] = lu[:n - 1, 2] * rb[1:]
    r = der2_solve(g.der2, lu, sl, dsdx)
    if dealiasing is not None:
        from .tlab_oracle_filter import opr_filter_1d
        vl, dsdx = opr_filter_1d(dealiasing, vl), opr_filter_1d(dealiasing, dsdx)      # uf, dsf (:479-482)
    if anelastic is not None and idir != 2:
        ri = np.asarray(anelastic[1], dtype=np.float64)
        nl = sl.shape[1]
        jy = (np.arange(nl) % ny) if idir == 1 else (np.arange(nl) // nx)       # ip of :149 / :179
        r = r * ri[jy][None, :] - vl * dsdx
    else:
        r = r - vl * dsdx
    return _from_lines(r, nx, ny, nz, idir), sl.ravel()


def boundary_bcs_neumann_y(ibc, nx, ny, nz, g, u):
    """tools/dns/boundary_bcs.f90:368-473 BOUNDARY_BCS_NEUMANN_Y (serial; C1N6, i.e. MatMul_5d_antisym with its bcs_b/bcs_t outputs,
    fdm_matmul.f90:384,410): wall values of u such that du/dy = 0 at the walls selected by ibc.  Returns (bcs_hb, bcs_ht) as (nz, nx)."""
    d = g.der1
    assert d.nb_diag == (3, 5), "oracle: BOUNDARY_BCS_NEUMANN_Y restated for the CompactJacobian6 first derivative"
    n = ny
    ul = _to_lines(np.asarray(u, dtype=np.float64), nx, ny, nz, 2)       # (ny, nx*nz), line = k + nz*i
    dst = np.empty_like(ul)
    ip = ibc * 5
    nmin, nmax = 0, n
    if ibc in (BCS_ND, BCS_NN):
        dst[0] = 0.0
        nmin += 1
    if ibc in (BCS_DN, BCS_NN):
        dst[n - 1] = 0.0
        nmax -= 1
    rb, rt = d.rhs_b, d.rhs_t
    hb = np.zeros(ul.shape[1])
    ht = np.zeros(ul.shape[1])
    if ibc in (BCS_ND, BCS_NN):
        hb = dst[0] * rb[0, 3] + ul[1] * rb[0, 4] + ul[2] * rb[0, 5] + ul[3] * rb[0, 1]
    matmul_5d_antisym(d.rhs, ul, dst, ibc, rb, rt)
    if ibc in (BCS_DN, BCS_NN):
        ht = ul[n - 4] * rt[3, 4] + ul[n - 3] * rt[3, 0] + ul[n - 2] * rt[3, 1] + dst[n - 1] * rt[3, 2]
    tridss(d.lu[nmin:nmax, ip], d.lu[nmin:nmax, ip + 1], d.lu[nmin:nmax, ip + 2], dst[nmin:nmax])
    if ibc in (BCS_ND, BCS_NN):
        hb = hb + d.lu[0, ip + 2] * dst[1]
    if ibc in (BCS_DN, BCS_NN):
        ht = ht + d.lu[n - 1, ip + 0] * dst[n - 2]
    # lines are (k fastest, then i): put the planes back in (nz, nx) order (x fastest)
    return (np.ascontiguousarray(hb.reshape(nx, nz).T), np.ascontiguousarray(ht.reshape(nx, nz).T))
