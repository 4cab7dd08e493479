"""CPU oracle, part 4: the 1-D filter kernels behind the dealiasing branch of OPR_Burgers_1D (physics/opr_burgers.f90:478-500) --
OPR_FILTER_1D (operators/opr_filter.f90:393-460) for the types COMPACT, 6E, 4E and COMPACT_CUTOFF.

TEST INFRASTRUCTURE ONLY.  numpy restatements of the right-hand sides / explicit stencils of src/filters/flt_compact.f90 and flt_explitic.f90
on top of the solvers of tlab_oracle.py; the coefficient tables f%coeffs are INPUT (OPR_FILTER_INITIALIZE stays the host's: its generators
FLT_C4_RHS_COEFFS / FLT_E4_COEFFS are not restated; the fixtures carry tables the reference made).  Pinned bitwise against the reference's
own modules through oracle/_ref (tests/test_oracle_filter.py, tests/golden/filters.npz).  The tophat family (flt_tophat.f90) is not restated."""
import numpy as np

from . import tlab_oracle as O

DNS_FILTER_NONE, DNS_FILTER_COMPACT, DNS_FILTER_6E, DNS_FILTER_4E, DNS_FILTER_TOPHAT, DNS_FILTER_COMPACT_CUTOFF = 0, 1, 2, 3, 8, 9
BCS_PERIODIC, BCS_BIASED, BCS_FREE, BCS_SOLID, BCS_DIRICHLET, BCS_NEUMANN, BCS_ZERO = range(7)     # filters/flt_base.f90:5-11

# constants of the cutoff filter, flt_compact.f90:11-16
C4_ALPHA, C4_BETA, C4_A, C4_BD2, C4_CD2, C4_DD2 = 0.6522474, 0.1702929, 0.9891856, 0.66059, 0.1666774, 0.679925e-3


def flt_c4_rhs(periodic, bcsmin, bcsmax, cxi, u):
    """flt_compact.f90:225-296 FLT_C4_RHS.  u: (n, nlines); cxi: (n, >=5)."""
    n = u.shape[0]
    r = np.empty_like(u)
    c = cxi
    if periodic:
        r[0] = c[0, 0] * u[n - 2] + c[0, 1] * u[n - 1] + c[0, 2] * u[0] + c[0, 3] * u[1] + c[0, 4] * u[2]
        r[1] = c[1, 0] * u[n - 1] + c[1, 1] * u[0] + c[1, 2] * u[1] + c[1, 3] * u[2] + c[1, 4] * u[3]
        r[n - 1] = c[n - 1, 0] * u[n - 3] + c[n - 1, 1] * u[n - 2] + c[n - 1, 2] * u[n - 1] + c[n - 1, 3] * u[0] + c[n - 1, 4] * u[1]
        r[n - 2] = c[n - 2, 0] * u[n - 4] + c[n - 2, 1] * u[n - 3] + c[n - 2, 2] * u[n - 2] + c[n - 2, 3] * u[n - 1] + c[n - 2, 4] * u[0]
    else:
        r[0] = c[0, 0] * u[0] + c[0, 1] * u[1] + c[0, 2] * u[2] + c[0, 3] * u[3] + c[0, 4] * u[4]
        r[1] = c[1, 0] * u[0] + c[1, 1] * u[1] + c[1, 2] * u[2] + c[1, 3] * u[3] + c[1, 4] * u[4]
        r[n - 2] = c[n - 2, 4] * u[n - 1] + c[n - 2, 3] * u[n - 2] + c[n - 2, 2] * u[n - 3] + c[n - 2, 1] * u[n - 4] + c[n - 2, 0] * u[n - 5]
        r[n - 1] = c[n - 1, 4] * u[n - 1] + c[n - 1, 3] * u[n - 2] + c[n - 1, 2] * u[n - 3] + c[n - 1, 1] * u[n - 4] + c[n - 1, 0] * u[n - 5]
        if bcsmin == BCS_ZERO:
            r[0] = u[0]
        if bcsmax == BCS_ZERO:
            r[n - 1] = u[n - 1]
    for i in range(2, n - 2):
        r[i] = c[i, 0] * u[i - 2] + c[i, 1] * u[i - 1] + c[i, 2] * u[i] + c[i, 3] * u[i + 1] + c[i, 4] * u[i + 2]
    return r


def flt_c4p_cutoff_rhs(u):
    """flt_compact.f90:327-349 FLT_C4P_CUTOFF_RHS (periodic)."""
    n = u.shape[0]
    r = np.empty_like(u)
    for i in range(n):
        r[i] = (C4_BD2 * (u[(i + 1) % n] + u[(i - 1) % n]) + C4_CD2 * (u[(i + 2) % n] + u[(i - 2) % n]) +
                C4_DD2 * (u[(i + 3) % n] + u[(i - 3) % n]) + C4_A * u[i])
    return r


def flt_c4_cutoff_rhs(u):
    """flt_compact.f90:351-375 FLT_C4_CUTOFF_RHS (biased ends)."""
    n = u.shape[0]
    r = np.empty_like(u)
    r[0] = (15.0 * u[0] + 4.0 * u[1] - 6.0 * u[2] + 4.0 * u[3] - u[4]) / 16.0
    r[1] = (12.0 * u[1] + u[0] + 6.0 * u[2] - 4.0 * u[3] + u[4]) / 16.0
    r[2] = (10.0 * u[2] - u[0] + 4.0 * u[1] + 4.0 * u[3] - u[4]) / 16.0
    r[n - 3] = (10.0 * u[n - 3] - u[n - 1] + 4.0 * u[n - 2] + 4.0 * u[n - 4] - u[n - 5]) / 16.0
    r[n - 2] = (12.0 * u[n - 2] + u[n - 1] + 6.0 * u[n - 3] - 4.0 * u[n - 4] + u[n - 5]) / 16.0
    r[n - 1] = (15.0 * u[n - 1] + 4.0 * u[n - 2] - 6.0 * u[n - 3] + 4.0 * u[n - 4] - u[n - 5]) / 16.0
    for i in range(3, n - 3):
        r[i] = C4_BD2 * (u[i + 1] + u[i - 1]) + C4_CD2 * (u[i + 2] + u[i - 2]) + C4_DD2 * (u[i + 3] + u[i - 3]) + C4_A * u[i]
    return r


_B = (11.0 / 16.0, 15.0 / 64.0, -3.0 / 32.0, 1.0 / 64.0)
_BB = (1.0 / 16.0, 3.0 / 4.0, 3.0 / 8.0, -1.0 / 4.0, 1.0 / 16.0, 0.0, 0.0)
_BC = (-1.0 / 32.0, 5.0 / 32.0, 11.0 / 16.0, 5.0 / 16.0, -5.0 / 32.0, 1.0 / 32.0, 0.0)


def flt_e6(periodic, bcs1, bcsn, u):
    """flt_explitic.f90:179-362 FLT_E6: explicit 6th-order filter; biased stencils on rows 2, 3 / n-1, n-2 when the end is BCS_BIASED (= 1),
    copies otherwise; the wall rows are copied."""
    n = u.shape[0]
    uf = np.empty_like(u)
    b0, b1, b2, b3 = _B
    ks, ke = 0, n
    if not periodic:
        uf[0] = u[0]
        if bcs1 == 1:
            k = 1
            uf[k] = (_BB[0] * u[k - 1] + _BB[1] * u[k] + _BB[2] * u[k + 1] + _BB[3] * u[k + 2] + _BB[4] * u[k + 3] + _BB[5] * u[k + 4] +
                     _BB[6] * u[k + 5])
            k = 2
            uf[k] = (_BC[0] * u[k - 2] + _BC[1] * u[k - 1] + _BC[2] * u[k] + _BC[3] * u[k + 1] + _BC[4] * u[k + 2] + _BC[5] * u[k + 3] +
                     _BC[6] * u[k + 4])
        else:
            uf[1], uf[2] = u[1], u[2]
        ks = 3
        uf[n - 1] = u[n - 1]
        if bcsn == 1:
            k = n - 2
            uf[k] = (_BB[0] * u[k + 1] + _BB[1] * u[k] + _BB[2] * u[k - 1] + _BB[3] * u[k - 2] + _BB[4] * u[k - 3] + _BB[5] * u[k - 4] +
                     _BB[6] * u[k - 5])
            k = n - 3
            uf[k] = (_BC[0] * u[k + 2] + _BC[1] * u[k + 1] + _BC[2] * u[k] + _BC[3] * u[k - 1] + _BC[4] * u[k - 2] + _BC[5] * u[k - 3] +
                     _BC[6] * u[k - 4])
        else:
            uf[n - 3], uf[n - 2] = u[n - 3], u[n - 2]
        ke = n - 3
    for k in range(ks, ke):
        uf[k] = (b3 * (u[(k - 3) % n] + u[(k + 3) % n]) + b2 * (u[(k - 2) % n] + u[(k + 2) % n]) + b1 * (u[(k - 1) % n] + u[(k + 1) % n]) +
                 b0 * u[k])
    return uf


def flt_e4(periodic, a, u):
    """flt_explitic.f90:17-62 FLT_E4 (coefficients a(n, 5) from FLT_E4_COEFFS)."""
    n = u.shape[0]
    uf = np.empty_like(u)
    if periodic:
        i = 0
        uf[i] = a[i, 0] * u[n - 2] + a[i, 1] * u[n - 1] + a[i, 2] * u[i] + a[i, 3] * u[i + 1] + a[i, 4] * u[i + 2]
        i = 1
        uf[i] = a[i, 0] * u[n - 1] + a[i, 1] * u[i - 1] + a[i, 2] * u[i] + a[i, 3] * u[i + 1] + a[i, 4] * u[i + 2]
        i = n - 2
        uf[i] = a[i, 0] * u[i - 2] + a[i, 1] * u[i - 1] + a[i, 2] * u[i] + a[i, 3] * u[i + 1] + a[i, 4] * u[0]
        i = n - 1
        uf[i] = a[i, 0] * u[i - 2] + a[i, 1] * u[i - 1] + a[i, 2] * u[i] + a[i, 3] * u[0] + a[i, 4] * u[1]
    else:
        i = 1
        uf[0] = u[0]
        uf[i] = a[i, 1] * u[i - 1] + a[i, 2] * u[i] + a[i, 3] * u[i + 1] + a[i, 4] * u[i + 2] + a[i, 0] * u[i + 3]
        i = n - 2
        uf[i] = a[i, 0] * u[i - 2] + a[i, 1] * u[i - 1] + a[i, 2] * u[i] + a[i, 3] * u[i + 1] + a[i, 4] * u[i - 3]
        uf[n - 1] = u[n - 1]
    for i in range(2, n - 2):
        uf[i] = a[i, 0] * u[i - 2] + a[i, 1] * u[i - 1] + a[i, 2] * u[i] + a[i, 3] * u[i + 1] + a[i, 4] * u[i + 2]
    return uf


class Filter:
    """type(filter_dt) as OPR_FILTER_INITIALIZE leaves it (opr_filter.f90:28-41): type, periodic, BcsMin/BcsMax, coeffs (n, inb_filter)."""

    def __init__(self, ftype, n, periodic, coeffs=None, bcsmin=BCS_BIASED, bcsmax=BCS_BIASED):
        self.type, self.size, self.periodic = int(ftype), int(n), bool(periodic)
        self.bcsmin, self.bcsmax = (BCS_PERIODIC, BCS_PERIODIC) if periodic else (int(bcsmin), int(bcsmax))
        self.coeffs = None if coeffs is None else np.array(coeffs, dtype=np.float64)


def opr_filter_1d(f, u):
    """operators/opr_filter.f90:393-460 OPR_FILTER_1D.  u: (n, nlines) -> filtered (n, nlines)."""
    c = f.coeffs
    if f.type == DNS_FILTER_COMPACT:
        r = flt_c4_rhs(f.periodic, f.bcsmin, f.bcsmax, c, u)
        if f.periodic:
            O.tridpss(c[:, 5], c[:, 6], c[:, 7], c[:, 8], c[:, 9], r)
        else:
            O.tridss(c[:, 5], c[:, 6], c[:, 7], r)
        return r
    if f.type == DNS_FILTER_COMPACT_CUTOFF:
        if f.periodic:
            r = flt_c4p_cutoff_rhs(u)
            O.pentadpss(*(c[:, k] for k in range(7)), r)
        else:
            r = flt_c4_cutoff_rhs(u)
            O.pentadss2(*(c[:, k] for k in range(5)), r)
        return r
    if f.type == DNS_FILTER_6E:
        return flt_e6(f.periodic, f.bcsmin, f.bcsmax, u)
    if f.type == DNS_FILTER_4E:
        return flt_e4(f.periodic, c, u)
    raise NotImplementedError("oracle: filter type %d" % f.type)


def opr_filter(nx, ny, nz, f, u):
    """operators/opr_filter.f90:283-392 OPR_FILTER, directional branch (:368-389): the x, then the y, then the z filter, each f[d].repeat times
    (None = DNS_FILTER_NONE).  u: flat field -> filtered flat field."""
    u = np.array(u, dtype=np.float64)
    for d in (1, 2, 3):
        fd = f[d - 1]
        if fd is None:
            continue
        for _ in range(getattr(fd, "repeat", 1)):
            u = O._from_lines(opr_filter_1d(fd, O._to_lines(u, nx, ny, nz, d)), nx, ny, nz, d)
    return u
