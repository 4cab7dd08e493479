"""RHS_GLOBAL_INCOMPRESSIBLE_1 + RK substep composed from the REFERENCE'S OWN COMPILED ROUTINES (oracle/_ref/libtlab_ref.so, or another build of
the same sources through TLAB_REF_LIB): every derivative, every Burgers operator, every per-mode solve of the pressure and BOUNDARY_BCS_NEUMANN_Y run
in the reference's Fortran; only the pointwise sums of the composition and the Fourier transforms (FFTW is not in the image: numpy.fft) are Python.

TEST INFRASTRUCTURE ONLY.  Two uses:
  * a second statement of the composed path next to the numpy oracle (oracle/tlab_oracle_rhs.py), whose class it extends -- the composition is the
    same code, the arithmetic is the reference's (tests/test_ref_composed.py holds the two together);
  * the reference-made yardstick of the composed path: the same inputs through two legitimate builds of the reference (with / without fused
    multiply-adds, oracle/Makefile targets `all` and `fma`) differ by rounding only, and by how much is what a device error may be compared with
    (tests/golden/make_golden_yardsticks.py -> tests/golden/yardsticks.json).

Restates: default schemes (CompactJacobian6 / CompactJacobian6Hyper with the wall closure the flang build reads, DESIGN.md section 2 defect 1),
no-slip or free-slip velocity walls, Dirichlet or Neumann scalars, remove_divergence on or off.  The library keeps its plans in module variables:
ONE grid per process.
"""
import numpy as np

from . import ref_lib as R
from . import tlab_oracle as O
from .tlab_oracle_rhs import DnsOracle

_GRID = [None]


class RefComposedDns(DnsOracle):
    def __init__(self, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=None):
        # hyper_bc1_ext: None = the reference's plans as it builds them; a number = the one wall-row entry of the default second derivative that the
        # reference reads out of bounds replaced in ITS plan (ref_driver.f90::ref_fdm_set_hyper_bc1_ext) -- the reference's routines on the closure 0.0
        super().__init__(x, y, z, nscal=nscal, visc=visc, schmidt=schmidt, yuniform=yuniform, hyper_bc1_ext=hyper_bc1_ext)      # the numpy plans: lambda of the modes, singular flags, norm
        key = (len(x), len(y), len(z), float(np.sum(x)), float(np.sum(y)), float(np.sum(z)), bool(yuniform), hyper_bc1_ext)
        if _GRID[0] is not None and _GRID[0] != key:
            raise RuntimeError("the reference library holds one grid per process")
        if _GRID[0] is None:
            R.init(self.nx, self.ny, self.nz)
            R.fdm_create(1, x, True, True)
            R.fdm_create(2, y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext)
            if self.nz > 1:
                R.fdm_create(3, z, True, True)
            _GRID[0] = key

    # ---- the reference's routines in place of the numpy restatements ----
    def burgers(self, d, nu, s, vel):
        if d == 3 and self.nz == 1:
            return np.zeros(self.n)                 # (as O.opr_burgers: no z dynamics in two dimensions)
        return R.burgers(d, self.nx, self.ny, self.nz, 0, nu, s, vel)[0]

    def p1(self, d, u):
        if d == 3 and self.nz == 1:
            return np.zeros(self.n)
        return R.partial(d, O.OPR_P1, self.nx, self.ny, self.nz, 0, u)[0]

    def neumann_y(self, ibc, a):
        hb, ht = R.bcs_neumann_y(ibc, self.nx, self.ny, self.nz, a)
        return hb.reshape(self.nz, self.nx), ht.reshape(self.nz, self.nx)

    def solve_poisson(self, f, hb, ht):
        """OPR_Poisson_FourierXZ_Factorize (opr_elliptic.f90:263-364): transforms by numpy.fft, the per-mode stage (:308-333) = FDM_Int1_Initialize +
        OPR_ODE2_Factorize_NN / _NN_Sing of the reference itself, one call per Fourier mode."""
        plan = self.poisson
        nx, ny, nz, nxh = self.nx, self.ny, self.nz, plan.nxh
        a = np.array(f, dtype=np.float64).reshape(nz, ny, nx).copy()
        a[:, 0, :] = np.asarray(hb).reshape(nz, nx)
        a[:, ny - 1, :] = np.asarray(ht).reshape(nz, nx)
        c = np.fft.rfft(a, axis=2)
        if nz > 1:
            c = np.fft.fft(c, axis=0)
        c = c * plan.norm
        M = nz * nxh
        lam = np.sqrt(plan.lam2.reshape(M))
        sing = plan.sing.reshape(M)
        u = np.zeros((ny, 2, M))
        v = np.zeros((ny, 2, M))
        cr = np.ascontiguousarray(c.real.transpose(0, 2, 1))        # (kz, kx, y)
        ci = np.ascontiguousarray(c.imag.transpose(0, 2, 1))
        fm = np.empty((ny, 2))
        bcs = np.empty((2, 2))
        for m in range(M):
            kz, kx = divmod(m, nxh)
            fm[:, 0] = cr[kz, kx]
            fm[:, 1] = ci[kz, kx]
            bcs[0] = fm[0]
            bcs[1] = fm[ny - 1]
            um, vm = R.ode2(2 if sing[m] else 1, lam[m], fm, bcs)
            u[:, :, m], v[:, :, m] = um, vm

        def back(w):
            cc = (w[:, 0, :] + 1j * w[:, 1, :]).reshape(ny, nz, nxh).transpose(1, 0, 2)
            if nz > 1:
                cc = np.fft.ifft(cc, axis=0) * nz
            return (np.fft.irfft(cc, n=nx, axis=2) * nx).reshape(-1)
        return back(u), back(v)
