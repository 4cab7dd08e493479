"""Host-side mirror of the reference's operator interface for the hot path (same names, argument order and
meaning as the Fortran module procedures), working on torch CUDA tensors whose storage is handed to the C ABI
as raw device pointers.  torch is plumbing here (HBM allocation, streams); all arithmetic is in the HIP kernels.

    OPR_Partial_X(type, nx, ny, nz, bcs, g, u, result, tmp1)        operators/opr_partial.f90:31
    OPR_Burgers_X(ivel, is, nx, ny, nz, bcs, s, u, result, tmp1, u_t) physics/opr_burgers.f90:190
    TLab_Transpose(a, nra, nca, ma, b, mb)                           utils/tlab_transpose.f90:14

Fields are flat fp64 tensors of nx*ny*nz elements, x fastest, exactly the reference's u(nx*ny*nz).
"""
import ctypes
import numpy as np

from .lib import load, check, TlabError, c_vp

# operators/opr_partial.f90:19-21, physics/opr_burgers.f90:29-30, base/tlab_constants.f90:63-66, fdm_derivative.f90:51-54
OPR_P1, OPR_P2, OPR_P2_P1 = 1, 2, 3
OPR_P1_INT_VP, OPR_P1_INT_PV, OPR_P0_INT_VP, OPR_P0_INT_PV = 5, 6, 7, 8      # interpolatory operators of the staggered pressure grid
OPR_B_SELF, OPR_B_U_IN = 0, 1
BCS_DD, BCS_ND, BCS_DN, BCS_NN = 0, 1, 2, 3
FDM_COM4_JACOBIAN, FDM_COM6_JACOBIAN_PENTA, FDM_COM6_JACOBIAN, FDM_COM6_JACOBIAN_HYPER = 4, 5, 6, 7
FDM_COM6_DIRECT, FDM_COM4_DIRECT = 16, 17

_initialised = False


def init(device=0):
    """tlab_init: selects the GPU.  Raises TlabError if no MI355X is visible."""
    global _initialised
    check(load().tlab_init(int(device)), "tlab_init")
    _initialised = True


def sync():
    check(load().tlab_sync(), "tlab_sync")


def _use_torch_stream():
    import torch
    load().tlab_set_stream(c_vp(torch.cuda.current_stream().cuda_stream))


def _ptr(t, n=None, name="tensor"):
    import torch
    if t is None:
        return c_vp(0)
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
        raise TlabError("%s must be a contiguous float64 CUDA tensor" % name)
    if n is not None and t.numel() < n:
        raise TlabError("%s has %d elements, needs %d" % (name, t.numel(), n))
    return c_vp(t.data_ptr())


class FdmPlan:
    """type(fdm_dt) (fdm/fdm.f90:14-29): compact-FDM plan of one direction, owned by the library.

    FdmPlan(nodes, periodic, uniform, ...) restates FDM_CreatePlan (fdm/fdm.f90:143-252).
    FdmPlan.from_arrays(...) takes the tables an unchanged Fortran host built in FDM_Initialize.
    hyper_bc1_ext: see include/tlab_amd.h (0.1 reproduces the flang-built reference)."""

    _KEYS = {"lhs1": (1, 5), "rhs1": (2, 7), "lu1": (3, None), "rhs_b1": (4, None), "rhs_t1": (5, None), "mwn1": (6, 1),
             "lhs2": (7, 5), "rhs2": (8, 12), "lu2": (9, None), "mwn2": (10, 1), "jac": (11, 3), "lu0i": (12, 5), "lu1i": (13, 5)}

    def __init__(self, nodes, periodic, uniform, scheme1=FDM_COM6_JACOBIAN, scheme2=FDM_COM6_JACOBIAN_HYPER,
                 hyper_bc1_ext=0.1, stagger=False):
        """stagger: [Staggering] StaggerHorizontalPressure (a periodic direction gets the interpolation tables and the interpolatory der1%mwn,
        fdm.f90:236-248); non-periodic directions ignore it, as the reference does."""
        nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        self.size = int(nodes.shape[0])
        self.periodic = bool(periodic)
        self.uniform = bool(uniform)
        self._h = c_vp(0)
        check(load().tlab_fdm_plan_create(ctypes.byref(self._h), self.size,
                                          nodes.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(self.periodic),
                                          int(self.uniform), int(scheme1), int(scheme2), float(hyper_bc1_ext)),
              "tlab_fdm_plan_create")
        if stagger and self.periodic:
            self.set_stagger(1)

    def set_stagger(self, mode):
        """tlab_fdm_plan_set_stagger: 1 tables + interpolatory wavenumbers, 2 tables only (host-built plans), 0 off"""
        check(load().tlab_fdm_plan_set_stagger(self._h, int(mode)), "tlab_fdm_plan_set_stagger")

    @classmethod
    def from_arrays(cls, n, periodic, need_1der, lhs1, rhs1, lhs2, rhs2, ndl1=3):
        """lhs*/rhs*: numpy arrays [row, diagonal] (as oracle / golden files hold them); rhs2 includes the 3
        Jacobian-correction columns after its ndr2 diagonals.  ndl1 = 5 with 7 rhs1 diagonals: CompactJacobian6Penta."""
        self = cls.__new__(cls)
        self.size, self.periodic, self.uniform = int(n), bool(periodic), not bool(need_1der)
        self._h = c_vp(0)
        ndr1, ndr2 = rhs1.shape[1], rhs2.shape[1] - 3
        f = [np.asfortranarray(a, dtype=np.float64) for a in (lhs1[:, :ndl1], rhs1, lhs2[:, :3], rhs2)]
        dp = ctypes.POINTER(ctypes.c_double)
        check(load().tlab_fdm_plan_create_from_arrays(ctypes.byref(self._h), self.size, int(periodic), int(need_1der),
                                                      int(ndl1), ndr1, f[0].ctypes.data_as(dp), f[1].ctypes.data_as(dp),
                                                      3, ndr2, f[2].ctypes.data_as(dp), f[3].ctypes.data_as(dp)),
              "tlab_fdm_plan_create_from_arrays")
        return self

    @classmethod
    def from_tables(cls, tab, periodic=False, scheme1=FDM_COM6_JACOBIAN, scheme2=FDM_COM6_JACOBIAN_HYPER):
        """Everything a Fortran host hands over for one direction: tab = dict with ndr1, ndr2, need_1der, lhs1, rhs1, lhs2, rhs2, mwn1,
        mwn2, jac (as oracle/ref_lib.fdm_arrays / the golden files hold them), plus the mode_fdm of both derivatives
        (FDM_COM6_DIRECT = 16 as scheme2: per-row right-hand side, SpaceOrder2 = CompactDirect6)."""
        rhs1 = np.asarray(tab["rhs1"])[:, :int(tab["ndr1"])]
        rhs2 = np.asarray(tab["rhs2"])[:, :int(tab["ndr2"]) + 3]
        self = cls.from_arrays(np.asarray(tab["lhs1"]).shape[0], periodic, int(tab["need_1der"]), np.asarray(tab["lhs1"]), rhs1,
                               np.asarray(tab["lhs2"]), rhs2)
        dp = ctypes.POINTER(ctypes.c_double)
        aux = [np.ascontiguousarray(tab[k], dtype=np.float64) if k in tab else None for k in ("mwn1", "mwn2")]
        jac = np.asfortranarray(tab["jac"], dtype=np.float64) if "jac" in tab else None
        nodes = np.ascontiguousarray(tab["nodes"], dtype=np.float64) if "nodes" in tab else None
        ptr = lambda a: a.ctypes.data_as(dp) if a is not None else None      # noqa: E731
        check(load().tlab_fdm_plan_set_aux(self._h, ctypes.cast(ptr(aux[0]), c_vp), ctypes.cast(ptr(aux[1]), c_vp),
                                           ctypes.cast(ptr(jac), c_vp), ctypes.cast(ptr(nodes), c_vp)), "tlab_fdm_plan_set_aux")
        check(load().tlab_fdm_plan_set_scheme(self._h, int(scheme1), int(scheme2)), "tlab_fdm_plan_set_scheme")
        return self

    def info(self, what):
        return load().tlab_fdm_plan_info(self._h, what)

    @property
    def need_1der(self):
        return bool(self.info(5))

    def table(self, key):
        """Plan table as a numpy array indexed [row, diagonal] like the oracle's."""
        which, cols = self._KEYS[key]
        n = self.size
        buf = np.zeros(n * 24 + 64)
        m = load().tlab_fdm_plan_get(self._h, which, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), buf.shape[0])
        if m < 0:
            raise TlabError("tlab_fdm_plan_get(%s) failed" % key)
        if key == "rhs_b1":
            return buf[:m].reshape(8, 4).T.copy()
        if key == "rhs_t1":
            return buf[:m].reshape(7, 5).T.copy()
        if cols == 1:
            return buf[:m].copy()
        return buf[:m].reshape(m // n, n).T.copy()

    def debug_host_chunked_solve(self, which, ibc, chunks, f):
        f = np.ascontiguousarray(f, dtype=np.float64).copy()
        check(load().tlab_debug_host_chunked_solve(self._h, which, ibc, chunks, f.ctypes.data_as(ctypes.POINTER(ctypes.c_double))),
              "tlab_debug_host_chunked_solve")
        return f

    def __del__(self):
        try:
            if self._h:
                load().tlab_fdm_plan_destroy(self._h)
        except Exception:
            pass


def _ibc(bcs):
    """ibc = bcs(1,1) + 2*bcs(2,1) (opr_partial.f90:91); bcs is the reference's 2x2 integer array (or an int ibc)."""
    if isinstance(bcs, (int, np.integer)):
        return int(bcs)
    b = np.asarray(bcs)
    return int(b[0, 0]) + 2 * int(b[1, 0])


def _partial(idir, type, nx, ny, nz, bcs, g, u, result, tmp1=None):
    n = nx * ny * nz
    _use_torch_stream()
    check(load().tlab_opr_partial(idir, g._h, int(type), nx, ny, nz, _ibc(bcs), _ptr(u, n, "u"), _ptr(result, n, "result"),
                                  _ptr(tmp1, n, "tmp1")), "tlab_opr_partial")


def OPR_Partial_X(type, nx, ny, nz, bcs, g, u, result, tmp1=None):
    _partial(1, type, nx, ny, nz, bcs, g, u, result, tmp1)


def OPR_Partial_Y(type, nx, ny, nz, bcs, g, u, result, tmp1=None):
    _partial(2, type, nx, ny, nz, bcs, g, u, result, tmp1)


def OPR_Partial_Z(type, nx, ny, nz, bcs, g, u, result, tmp1=None):
    _partial(3, type, nx, ny, nz, bcs, g, u, result, tmp1)


def BOUNDARY_BCS_NEUMANN_Y(ibc, nx, ny, nz, g, u, bcs_hb, bcs_ht, tmp1):
    """tools/dns/boundary_bcs.f90:368: wall planes (nx*nz) of u such that du/dy = 0 at jmin (ibc=1), jmax (2) or both (3)."""
    n = nx * ny * nz
    _use_torch_stream()
    check(load().tlab_boundary_bcs_neumann_y(g._h, int(ibc), nx, ny, nz, _ptr(u, n, "u"), _ptr(bcs_hb, nx * nz, "bcs_hb"),
                                             _ptr(bcs_ht, nx * nz, "bcs_ht"), _ptr(tmp1, n, "tmp1")), "tlab_boundary_bcs_neumann_y")


def _burgers(idir, ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, write_transposed):
    n = nx * ny * nz
    b = np.asarray(bcs) if not isinstance(bcs, (int, np.integer)) else None
    if b is not None and b.shape == (2, 2) and int(b[0, 1]) + int(b[1, 1]) > 0:
        raise TlabError("OPR_Burgers: only developed for biased BCs (opr_burgers.f90:460-463)")
    _use_torch_stream()
    check(load().tlab_opr_burgers(idir, g._h, int(ivel), nx, ny, nz, _ibc(bcs), float(nu), _ptr(s, n, "s"), _ptr(u, n, "u"),
                                  _ptr(result, n, "result"), _ptr(tmp1, n, "tmp1"), int(write_transposed)),
          "tlab_opr_burgers")


# `is` (scalar index selecting the diffusivity in the reference's module state) becomes the diffusivity itself:
# nu = visc for is = 0, visc/schmidt(is) otherwise (opr_burgers.f90:94-98).  u_t is accepted and ignored (see tlab_amd.h).
def OPR_Burgers_X(ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, u_t=None, write_transposed=False):
    _burgers(1, ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, write_transposed)


def OPR_Burgers_Y(ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, u_t=None, write_transposed=False):
    _burgers(2, ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, write_transposed)


def OPR_Burgers_Z(ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, u_t=None, write_transposed=False):
    _burgers(3, ivel, nu, nx, ny, nz, bcs, g, s, u, result, tmp1, write_transposed)


class Filter:
    """type(filter_dt) of operators/opr_filter.f90:28-41 with the coefficient table OPR_FILTER_INITIALIZE made on the host: ftype = 1 compact,
    2 explicit6, 3 explicit4, 9 compactcutoff; coeffs: numpy [row, column] (n, inb_filter) or None (explicit6); bcsmin / bcsmax: DNS_FILTER_BCS_*
    (filters/flt_base.f90: 1 biased, 2 free, 6 zero ...)."""

    def __init__(self, ftype, n, periodic, coeffs=None, bcsmin=1, bcsmax=1):
        self.type, self.size, self.periodic = int(ftype), int(n), bool(periodic)
        self._h = c_vp(0)
        dp = ctypes.POINTER(ctypes.c_double)
        if coeffs is None:
            nc, ptr = 0, None
        else:
            self._c = np.asfortranarray(coeffs, dtype=np.float64)
            nc, ptr = self._c.shape[1], self._c.ctypes.data_as(dp)
        check(load().tlab_filter_create(ctypes.byref(self._h), self.type, self.size, int(self.periodic), 0 if periodic else int(bcsmin),
                                        0 if periodic else int(bcsmax), nc, ptr), "tlab_filter_create")

    def __del__(self):
        try:
            if self._h:
                load().tlab_filter_destroy(self._h)
        except Exception:
            pass


def OPR_FILTER_1D(idir, f, nx, ny, nz, u, result):
    """OPR_FILTER_1D (opr_filter.f90:393-460) along direction idir of a field, out of place."""
    _use_torch_stream()
    check(load().tlab_opr_filter_1d(int(idir), f._h, nx, ny, nz, _ptr(u), _ptr(result)), "tlab_opr_filter_1d")


def set_dealiasing(idir, f):
    """Dealiasing(idir) of OPR_Burgers ([Dealiasing], opr_burgers.f90:33, 71): a Filter, or None for DNS_FILTER_NONE.  Module state of the
    operator, as in the reference; the caller keeps the Filter alive while it is set."""
    check(load().tlab_opr_burgers_set_dealiasing(int(idir), f._h if f is not None else None), "tlab_opr_burgers_set_dealiasing")


class PoissonPlan:
    """Module state of OPR_Elliptic (operators/opr_elliptic.f90:64-81) + OPR_Fourier plans, built by
    OPR_Elliptic_Initialize / OPR_Fourier_Initialize in the reference; here one object owned by the library."""

    def __init__(self, gx, gy, gz, nx, ny, nz, gy_elliptic=None):
        """gy_elliptic: the reference's fdm_loc of EllipticOrder = CompactDirect6 (a y plan with the direct second derivative and its nodes,
        FdmPlan.from_tables) -> OPR_Poisson_FourierXZ_Direct; None -> the factorized solver built from gy%der1."""
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.isize_txc_field = (self.nx + 2) * self.ny * self.nz     # base/tlab_memory.f90:186-187
        self._keep = (gx, gy, gz, gy_elliptic)
        self._h = c_vp(0)
        self.direct = gy_elliptic is not None
        if self.direct:
            check(load().tlab_poisson_plan_create_direct(ctypes.byref(self._h), gx._h, gy._h, gz._h, self.nx, self.ny, self.nz, gy_elliptic._h),
                  "tlab_poisson_plan_create_direct")
        else:
            check(load().tlab_poisson_plan_create(ctypes.byref(self._h), gx._h, gy._h, gz._h, self.nx, self.ny, self.nz),
                  "tlab_poisson_plan_create")

    def __del__(self):
        try:
            if self._h:
                load().tlab_poisson_plan_destroy(self._h)
        except Exception:
            pass


def OPR_Poisson(plan, nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy=None):
    """OPR_Poisson(nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)  operators/opr_elliptic.f90:33-46.
    `plan` replaces the reference's module state.  tmp1, tmp2 need (nx+2)*ny*nz elements."""
    n = nx * ny * nz
    _use_torch_stream()
    check(load().tlab_opr_poisson(plan._h, nx, ny, nz, int(ibc), _ptr(p, n, "p"), _ptr(tmp1, plan.isize_txc_field, "tmp1"),
                                  _ptr(tmp2, plan.isize_txc_field, "tmp2"), _ptr(bcs_hb, nx * nz, "bcs_hb"),
                                  _ptr(bcs_ht, nx * nz, "bcs_ht"), _ptr(dpdy, n, "dpdy")), "tlab_opr_poisson")


def poisson_set_exact(on):
    """Factorized Poisson plans created after this call use the marching per-mode solver that repeats the reference's operations one by one
    (bit-faithful up to the FFTs, 2x the time of the per-mode stage) instead of the register-chunked one; include/tlab_amd.h."""
    check(load().tlab_poisson_set_exact(int(bool(on))), "tlab_poisson_set_exact")


def OPR_Helmholtz(plan, nx, ny, nz, ibc, alpha, a, tmp1, tmp2, bcs_hb, bcs_ht):
    """OPR_Helmholtz(nx, ny, nz, ibc, alpha, a, tmp1, tmp2, bcs_hb, bcs_ht)  operators/opr_elliptic.f90:48-62: lap a + alpha a = f.
    Direct plan (PoissonPlan(..., gy_elliptic=...)): OPR_Helmholtz_FourierXZ_Direct :562-628, any boundary type; factorized plan:
    OPR_Helmholtz_FourierXZ_Factorize :466-557, BCS_NN or BCS_DD (the tables of an alpha are built on its first call)."""
    n = nx * ny * nz
    _use_torch_stream()
    check(load().tlab_opr_helmholtz(plan._h, nx, ny, nz, int(ibc), float(alpha), _ptr(a, n, "a"), _ptr(tmp1, plan.isize_txc_field, "tmp1"),
                                    _ptr(tmp2, plan.isize_txc_field, "tmp2"), _ptr(bcs_hb, nx * nz, "bcs_hb"), _ptr(bcs_ht, nx * nz, "bcs_ht")),
          "tlab_opr_helmholtz")


def TLab_Transpose(a, nra, nca, b):
    """b(nca, nra) = transpose of Fortran a(nra, nca); bit-exact."""
    _use_torch_stream()
    check(load().tlab_transpose(_ptr(a, nra * nca, "a"), nra, nca, _ptr(b, nra * nca, "b")), "tlab_transpose")
