"""CPU oracle, part 3: numpy restatement of RHS_GLOBAL_INCOMPRESSIBLE_1 and of the explicit RK substep, composed from the
operator oracles (tlab_oracle.py, tlab_oracle_poisson.py) in the reference's call order.

TEST INFRASTRUCTURE ONLY.  Parity status: every operator it composes is pinned against the reference (see the two
modules); the composition itself follows tools/dns/rhs_global_incompressible_1.f90:98-375 line by line (convective form,
RhsMode = combined, remove_divergence; wall BCs Dirichlet or Neumann per field, :360-398) and tools/dns/time.f90:645-664, :261-298.
The reference's driver (dns.x) cannot be built in this image (needs fftw3.f03 through opr_fourier.f90), so this level is
pinned through its parts plus the discrete invariant it must satisfy: div(q/dte + hq) = 0 in the interior."""
import numpy as np

from . import tlab_oracle as O
from . import tlab_oracle_poisson as OP


class DnsOracle:
    def __init__(self, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, plans=None, gy_elliptic=None, hyper_bc1_ext=None,
                 anelastic=None, dealiasing=None, stagger=False):
        """anelastic = (rbackground, ribackground): nse_eqns == DNS_EQNS_ANELASTIC with those background profiles (ny values each): the density
        weights of rhs_global_incompressible_1.f90:211-214, :275-277, :326-329 and of OPR_Burgers (opr_burgers.f90:128-183).  The buoyancy and
        the thermodynamics that PRODUCE the profiles are outside the path."""
        self.anelastic = None if anelastic is None else tuple(np.asarray(a, dtype=np.float64) for a in anelastic)
        self.dealiasing = [None, None, None] if dealiasing is None else list(dealiasing)      # Dealiasing(1:3) of opr_burgers.f90:33 (None: DNS_FILTER_NONE)
        self.nx, self.ny, self.nz = len(x), len(y), len(z)
        self.n = self.nx * self.ny * self.nz
        h = hyper_bc1_ext                      # None: O.HYPER_BC1_EXT = the flang-built reference's wall closure (defect 1)
        self.stagger = bool(stagger)                # [Staggering] StaggerHorizontalPressure = yes (TLab_WorkFlow::stagger_on)
        self.g = list(plans) if plans is not None else [O.FdmPlan(x, True, True, hyper_bc1_ext=h, stagger=stagger), O.FdmPlan(y, False, yuniform, hyper_bc1_ext=h),
                                                        O.FdmPlan(z, True, True, hyper_bc1_ext=h, stagger=stagger)]
        self.direct = gy_elliptic is not None         # EllipticOrder = CompactDirect6: OPR_Poisson => OPR_Poisson_FourierXZ_Direct (opr_elliptic.f90:153)
        if self.direct:
            self.poisson = OP.PoissonDirectPlan(self.g[0], gy_elliptic, self.g[2], self.nx, self.ny, self.nz)
        else:
            self.poisson = OP.PoissonPlan(self.g[0], self.g[1], self.g[2], self.nx, self.ny, self.nz, stagger=self.stagger)
        self.nscal, self.visc, self.schmidt = nscal, visc, list(schmidt)
        self.q = [np.zeros(self.n) for _ in range(3)]
        self.s = [np.zeros(self.n) for _ in range(nscal)]
        self.hq = [np.zeros(self.n) for _ in range(3)]
        self.hs = [np.zeros(self.n) for _ in range(nscal)]
        # BcsFlowJmin/Jmax%type, BcsScalJmin/Jmax%type (boundary_bcs.f90:18-27): 3 = DNS_BCS_DIRICHLET, 4 = DNS_BCS_NEUMANN
        self.flow_jmin, self.flow_jmax = [3, 3, 3], [3, 3, 3]
        self.scal_jmin, self.scal_jmax = [3] * nscal, [3] * nscal
        # BcsScalJmin/Jmax%SfcType (0 = DNS_SFC_STATIC, 1 = DNS_SFC_LINEAR) and %cpl (boundary_bcs.f90:29-31, 49-50, 76-87): dynamic surface model
        self.pressure_filter = [None, None, None]      # PressureFilter(1:3) (opr_filter.f90:46; rhs_global_incompressible_1.f90:286-290): tlab_oracle_filter.Filter or None
        self.remove_divergence = True                  # dns.ini; False: the else-branch of :234-250 (forcing = div(hq))
        self.sfc_jmin, self.sfc_jmax = [0] * nscal, [0] * nscal
        self.cpl_jmin, self.cpl_jmax = [0.0] * nscal, [0.0] * nscal

    def burgers(self, d, nu, s, vel):
        return O.opr_burgers(d, self.nx, self.ny, self.nz, 0, self.g[d - 1], nu, s, vel, anelastic=self.anelastic, dealiasing=self.dealiasing[d - 1])[0]

    def weight(self, w, a):
        """Thermo_Anelastic_WEIGHT_* (thermodynamics/thermo_anelastic.f90:377-448): a(i, j, k) * w(j)"""
        return (a.reshape(self.nz, self.ny, self.nx) * w[None, :, None]).ravel()

    def p1(self, d, u):
        return O.opr_partial(d, O.OPR_P1, self.nx, self.ny, self.nz, 0, self.g[d - 1], u)[0]

    def solve_poisson(self, f, hb, ht):
        """OPR_Poisson (factorized); a subclass may put the reference's compiled per-mode routines here (oracle/tlab_ref_rhs.py)"""
        return OP.opr_poisson_fxz(self.poisson, f, hb, ht)

    def neumann_y(self, ibc, a):
        """BOUNDARY_BCS_NEUMANN_Y -> (bcs_hb, bcs_ht)"""
        return O.boundary_bcs_neumann_y(ibc, self.nx, self.ny, self.nz, self.g[1], a)

    def pint(self, d, itype, u):
        """OPR_Partial_X / _Z with an interpolatory type (OPR_P0_INT_VP ...), opr_partial.f90:110-120, :228-238"""
        return O.opr_partial(d, itype, self.nx, self.ny, self.nz, 0, self.g[d - 1], u)[0]

    def rhs_global_incompressible_1(self, dte):
        nx, ny, nz = self.nx, self.ny, self.nz
        u, v, w = self.q
        hq, hs = self.hq, self.hs
        nu = self.visc
        # the old tendency of the scalar at the boundary for the dynamic surface BCs (:77-87); zero otherwise
        sref_b = [hs[i].reshape(nz, ny, nx)[:, 0, :].copy() if self.sfc_jmin[i] == 1 else np.zeros((nz, nx)) for i in range(self.nscal)]
        sref_t = [hs[i].reshape(nz, ny, nx)[:, ny - 1, :].copy() if self.sfc_jmax[i] == 1 else np.zeros((nz, nx)) for i in range(self.nscal)]
        tmp1 = self.burgers(1, nu, u, u); tmp2 = self.burgers(2, nu, v, v); tmp3 = self.burgers(3, nu, w, w)      # :98-100
        tmp7 = self.burgers(2, nu, u, v); tmp8 = self.burgers(3, nu, u, w)                                          # :103-104
        hq[0] = hq[0] + tmp1 + tmp7 + tmp8
        tmp7 = self.burgers(1, nu, v, u); tmp8 = self.burgers(3, nu, v, w)                                          # :115-116
        hq[1] = hq[1] + tmp2 + tmp7 + tmp8
        tmp7 = self.burgers(1, nu, w, u); tmp8 = self.burgers(2, nu, w, v)                                          # :127-128
        hq[2] = hq[2] + tmp3 + tmp7 + tmp8
        for i in range(self.nscal):                                                                                # :149-162
            kap = self.visc / self.schmidt[i]
            t1 = self.burgers(1, kap, self.s[i], u); t2 = self.burgers(2, kap, self.s[i], v); t3 = self.burgers(3, kap, self.s[i], w)
            hs[i] = hs[i] + t1 + t2 + t3
        if self.remove_divergence:
            dummy = 1.0 / dte                                                                                       # :188-201
            tmp2 = hq[1] + v * dummy
            tmp3 = hq[0] + u * dummy
            tmp4 = hq[2] + w * dummy
        else:                                                                                                       # :234-250
            tmp2, tmp3, tmp4 = hq[1].copy(), hq[0].copy(), hq[2].copy()
        if self.anelastic is not None:                                                                              # :211-214
            rb, ri = self.anelastic
            tmp2, tmp3, tmp4 = self.weight(rb, tmp2), self.weight(rb, tmp3), self.weight(rb, tmp4)
        VP0, VP1, PV0, PV1 = O.OPR_P0_INT_VP, O.OPR_P1_INT_VP, O.OPR_P0_INT_PV, O.OPR_P1_INT_PV
        if self.stagger:                                                                                            # :216-226: derivatives onto the pressure nodes
            tmp1 = self.pint(3, VP0, self.p1(2, self.pint(1, VP0, tmp2)))
            tmp2 = self.pint(3, VP0, self.pint(1, VP1, tmp3))
            tmp3 = self.pint(3, VP1, self.pint(1, VP0, tmp4))
        else:
            tmp1 = self.p1(2, tmp2); tmp2 = self.p1(1, tmp3); tmp3 = self.p1(3, tmp4)                               # :228-230
        tmp1 = tmp1 + tmp2 + tmp3                                                                                   # :258
        h2 = (self.pint(3, VP0, self.pint(1, VP0, hq[1])) if self.stagger else hq[1]).reshape(nz, ny, nx)           # :266-273
        hb, ht = h2[:, 0, :].copy(), h2[:, ny - 1, :].copy()                                                        # :279-280
        if self.anelastic is not None:                                                                              # :275-277
            hb, ht = hb * rb[0], ht * rb[ny - 1]
        if self.direct:
            p, dpdy = OP.opr_poisson_fxz_direct(self.poisson, tmp1, hb, ht, gy_der=self.g[1])
        else:
            p, dpdy = self.solve_poisson(tmp1, hb, ht)                                                              # :284
        if any(f is not None for f in self.pressure_filter):                                                       # :286-290
            from .tlab_oracle_filter import opr_filter
            p = opr_filter(nx, ny, nz, self.pressure_filter, p)
            dpdy = opr_filter(nx, ny, nz, self.pressure_filter, dpdy)
        self.p = p
        if self.stagger:                                                                                            # :307-317: back onto the velocity nodes
            dpdy = self.pint(1, PV0, self.pint(3, PV0, dpdy))
            tmp4 = self.pint(1, PV0, self.pint(3, PV1, p))
            tmp2 = self.pint(1, PV1, self.pint(3, PV0, p))
        else:
            tmp2 = self.p1(1, p); tmp4 = self.p1(3, p)                                                              # :319-320
        if self.anelastic is not None:                                                                              # :326-329
            hq[0] = hq[0] - self.weight(ri, tmp2); hq[1] = hq[1] - self.weight(ri, dpdy); hq[2] = hq[2] - self.weight(ri, tmp4)
        else:
            hq[0] = hq[0] - tmp2; hq[1] = hq[1] - dpdy; hq[2] = hq[2] - tmp4                                        # :349-351
        types = list(zip(self.flow_jmin, self.flow_jmax)) + list(zip(self.scal_jmin, self.scal_jmax))
        for ia, (a, (tmin, tmax)) in enumerate(zip(hq + hs, types)):                                                # :363-375, :379-396
            ref_b = np.zeros((nz, nx)); ref_t = np.zeros((nz, nx))
            if ia >= 3:
                ref_b, ref_t = sref_b[ia - 3], sref_t[ia - 3]
            ibc = (1 if tmin == 4 else 0) + (2 if tmax == 4 else 0)
            if ibc > 0:
                nb, nt = self.neumann_y(ibc, a)
                if ibc & 1:
                    ref_b = nb
                if ibc & 2:
                    ref_t = nt
            if ia >= 3 and (self.sfc_jmin[ia - 3] == 1 or self.sfc_jmax[ia - 3] == 1):                              # BOUNDARY_BCS_SURFACE_Y (boundary_bcs.f90:478-546)
                i = ia - 3
                diff = self.visc / self.schmidt[i]
                t1 = self.p1(2, self.s[i]).reshape(nz, ny, nx)                                                      # :508
                avg1 = self._avg1v2d(t1, 0)                                                                         # AVG1V2D(.., 1, 1, tmp1): j = 1 at BOTH ends (:520, :535)
                if self.sfc_jmin[i] == 1:
                    hfx = diff * t1[:, 0, :]
                    ref_b = ref_b + self.cpl_jmin[i] * (hfx - diff * avg1)
                if self.sfc_jmax[i] == 1:
                    hfx = -diff * t1[:, ny - 1, :]
                    ref_t = ref_t + self.cpl_jmax[i] * (hfx - diff * avg1)
            b = a.reshape(nz, ny, nx)
            b[:, 0, :] = ref_b
            b[:, ny - 1, :] = ref_t

    @staticmethod
    def _avg1v2d(a3, j):
        """utils/averages.f90:114-137 AVG1V2D (imom = 1): serial sum over i fastest, then k, of the plane j, / (nx nz)"""
        acc = 0.0
        for row in a3[:, j, :]:
            for v in row:
                acc = acc + v
        return acc / float(a3.shape[0] * a3.shape[2])

    def time_courant(self, cfla, cfld):
        """tools/dns/time.f90:365-548 (incompressible) with the constants of TIME_INITIALIZE :138-176."""
        nx, ny, nz = self.nx, self.ny, self.nz
        o1 = [1.0 / g.jac[:, 0] for g in self.g]
        u, v, w = (a.reshape(nz, ny, nx) for a in self.q)
        f = np.abs(u) * o1[0][None, None, :] + np.abs(v) * o1[1][None, :, None]
        if nz > 1:
            f = f + np.abs(w) * o1[2][:, None, None]
        pmax1 = f.max()
        sf = 1.0
        for sc in self.schmidt[:self.nscal]:
            sf = max(sf, 1.0 / sc)
        sf = sf * self.visc
        dx2i = sum((o * o).max() for o, n in zip(o1, (nx, ny, nz)) if n > 1)
        pmax2 = sf * dx2i
        dtc = cfla / pmax1 if pmax1 > 0 else 1e300
        dtd = cfld / pmax2 if pmax2 > 0 else 1e300
        return (pmax1, pmax2), min(dtc, dtd)

    def fi_invariant_p(self):
        """mappings/fi_vectorcalculus.f90:111-141."""
        r = self.p1(1, self.q[0])
        r = r + self.p1(2, self.q[1])
        return -(r + self.p1(3, self.q[2]))

    def time_substep(self, dte, kco=1.0, scale=False):
        self.rhs_global_incompressible_1(dte)
        for i in range(3):
            self.q[i] = self.q[i] + dte * self.hq[i]                                                                # time.f90:651
        for i in range(self.nscal):
            self.s[i] = self.s[i] + dte * self.hs[i]
        if scale:
            for i in range(3):
                self.hq[i] = kco * self.hq[i]                                                                       # time.f90:283
            for i in range(self.nscal):
                self.hs[i] = kco * self.hs[i]
