"""Host-side mirror of the RK driver around the hot path: module arrays q, s, hq, hs, txc (base/tlab_memory.f90:10-17),
RHS_GLOBAL_INCOMPRESSIBLE_1 (tools/dns/rhs_global_incompressible_1.f90:15), TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT and
TIME_RUNGEKUTTA (tools/dns/time.f90:559, :185).  Fields are torch CUDA tensors (HBM residency); every arithmetic
operation runs in the HIP library."""
import ctypes
import numpy as np

from .lib import load, check, TlabError, c_vp
from .operators import FdmPlan, PoissonPlan, _use_torch_stream

RKM_EXP3, RKM_EXP4 = 3, 4
DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN = 3, 4          # tools/dns/boundary_bcs.f90:18-19


def velocity_bcs(kind):
    """[BoundaryConditions] VelocityJmin/Jmax keyword -> BcsFlowJm%type(1:3), tools/dns/boundary_bcs.f90:112-121."""
    kind = kind.strip().lower()
    if kind == "noslip":
        return [DNS_BCS_DIRICHLET] * 3
    if kind == "freeslip":
        return [DNS_BCS_NEUMANN, DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN]
    raise TlabError("BoundaryConditions.Velocity: noslip or freeslip")


def scalar_bcs(kind):
    kind = kind.strip().lower()
    if kind == "dirichlet":
        return DNS_BCS_DIRICHLET
    if kind == "neumann":
        return DNS_BCS_NEUMANN
    raise TlabError("BoundaryConditions.Scalar: dirichlet or neumann")


def _bcs_arrays(nscal, velocity_jmin, velocity_jmax, scalar_jmin, scalar_jmax):
    IA = ctypes.c_int * 3
    fj0, fj1 = IA(*velocity_bcs(velocity_jmin)), IA(*velocity_bcs(velocity_jmax))
    SA = ctypes.c_int * max(nscal, 1)
    as_list = lambda v: [v] * nscal if isinstance(v, str) else list(v)      # noqa: E731
    sj0 = SA(*[scalar_bcs(k) for k in as_list(scalar_jmin)] or [DNS_BCS_DIRICHLET])
    sj1 = SA(*[scalar_bcs(k) for k in as_list(scalar_jmax)] or [DNS_BCS_DIRICHLET])
    return fj0, fj1, sj0, sj1


def rk_coefficients(mode):
    """TIME_INITIALIZE, tools/dns/time.f90:86-108."""
    if mode == RKM_EXP3:   # Williamson 1980
        return ([1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0], [-5.0 / 9.0, -153.0 / 128.0])
    if mode == RKM_EXP4:   # Carpenter & Kennedy 1994
        kdt = [1432997174477.0 / 9575080441755.0, 5161836677717.0 / 13612068292357.0, 1720146321549.0 / 2090206949498.0,
               3134564353537.0 / 4481467310338.0, 2277821191437.0 / 14882151754819.0]
        kco = [-567301805773.0 / 1357537059087.0, -2404267990393.0 / 2016746695238.0, -3550918686646.0 / 2091501179385.0,
               -1275806237668.0 / 842570457699.0]
        return (kdt, kco)
    raise TlabError("only the explicit low-storage schemes RungeKuttaExplicit3/4 are built")


class Dns:
    """imax, jmax, kmax, inb_scal, visc, schmidt + the allocated arrays of TLab_Initialize_Memory (tlab_memory.f90:164-216)."""

    def __init__(self, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3,
                 hyper_bc1_ext=0.0, device="cuda", plans=None, gy_elliptic=None, stagger=False):
        import torch
        self.nx, self.ny, self.nz = len(x), len(y), len(z)
        self.n = self.nx * self.ny * self.nz
        self.nscal = int(nscal)
        self.visc = float(visc)
        self.schmidt = np.ascontiguousarray(schmidt, dtype=np.float64)[: self.nscal]
        # plans: optional (gx, gy, gz) built elsewhere, e.g. FdmPlan.from_tables with a host's CompactDirect6 tables in y
        self.g = list(plans) if plans is not None else [
            FdmPlan(x, True, True, hyper_bc1_ext=hyper_bc1_ext, stagger=stagger), FdmPlan(y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext),
            FdmPlan(z, True, True, hyper_bc1_ext=hyper_bc1_ext, stagger=stagger)]
        # stagger: [Staggering] StaggerHorizontalPressure -- the driver reads it off the x plan (tlab_fdm_plan_info 7)
        # gy_elliptic: the y plan of EllipticOrder = CompactDirect6 (fdm_loc, opr_elliptic.f90:107-124) -> OPR_Poisson_FourierXZ_Direct
        self.poisson = PoissonPlan(self.g[0], self.g[1], self.g[2], self.nx, self.ny, self.nz, gy_elliptic=gy_elliptic)
        self.isize_txc_field = self.poisson.isize_txc_field
        f = lambda m: [torch.zeros(m, dtype=torch.float64, device=device) for _ in range(1)][0]   # noqa: E731
        self.q = [f(self.n) for _ in range(3)]
        self.s = [f(self.n) for _ in range(self.nscal)]
        self.hq = [f(self.n) for _ in range(3)]
        self.hs = [f(self.n) for _ in range(self.nscal)]
        self.txc = [f(self.isize_txc_field) for _ in range(9)]          # inb_txc = 9 (dns_read_local.f90:711)
        self.kdt, self.kco = rk_coefficients(rkm_mode)
        self.rkm_endstep = len(self.kdt)
        self._h = c_vp(0)
        sc = self.schmidt if self.nscal else np.zeros(1)
        check(load().tlab_dns_create(ctypes.byref(self._h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.poisson._h,
                                     self.nx, self.ny, self.nz, self.nscal, self.visc,
                                     sc.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), "tlab_dns_create")
        self._ptrs = None

    def set_bcs(self, velocity_jmin="noslip", velocity_jmax="noslip", scalar_jmin="dirichlet", scalar_jmax="dirichlet"):
        """Wall boundary conditions in y by the reference's dns.ini keywords ([BoundaryConditions], boundary_bcs.f90:102-190)."""
        fj0, fj1, sj0, sj1 = _bcs_arrays(self.nscal, velocity_jmin, velocity_jmax, scalar_jmin, scalar_jmax)
        check(load().tlab_dns_set_bcs(self._h, fj0, fj1, sj0, sj1), "tlab_dns_set_bcs")

    def set_pressure_filter(self, fx=None, fy=None, fz=None, repeat=None):
        """[PressureFilter]: tlab_amd.Filter objects per direction (None = no filter); the caller keeps them alive."""
        self._pfilters = (fx, fy, fz)
        rp = (ctypes.c_int * 3)(*(repeat or (1, 1, 1)))
        h = [f._h if f is not None else None for f in (fx, fy, fz)]
        check(load().tlab_dns_set_pressure_filter(self._h, h[0], h[1], h[2], rp), "tlab_dns_set_pressure_filter")

    def set_remove_divergence(self, on):
        """dns.ini remove_divergence (default on): forcing = div(hq + q/dte); off: div(hq)."""
        check(load().tlab_dns_set_remove_divergence(self._h, int(bool(on))), "tlab_dns_set_remove_divergence")

    def set_surface_bcs(self, sfc_jmin=None, sfc_jmax=None, coupling_jmin=None, coupling_jmax=None):
        """[BoundaryConditions] Scalar<i>SfcTypeJmin/Jmax = "static" | "linear" and Scalar<i>CouplingJmin/Jmax per scalar (boundary_bcs.f90:76-87)."""
        ns = max(self.nscal, 1)
        code = lambda v: [1 if str(t).lower() == "linear" else 0 for t in (v or ["static"] * ns)]      # noqa: E731
        s0, s1 = (ctypes.c_int * ns)(*code(sfc_jmin)[:ns]), (ctypes.c_int * ns)(*code(sfc_jmax)[:ns])
        c0 = (ctypes.c_double * ns)(*[float(v) for v in (coupling_jmin or [0.0] * ns)][:ns])
        c1 = (ctypes.c_double * ns)(*[float(v) for v in (coupling_jmax or [0.0] * ns)][:ns])
        check(load().tlab_dns_set_surface_bcs(self._h, s0, s1, c0, c1), "tlab_dns_set_surface_bcs")

    def set_anelastic(self, rbackground=None, ribackground=None):
        """nse_eqns = anelastic with the background density profile rbackground(ny) (ribackground defaults to 1 / rbackground); None: incompressible.
        Module state of the Burgers operator, like the reference's rhoinv: it applies to every plan of the process until switched off."""
        dp = ctypes.POINTER(ctypes.c_double)
        if rbackground is None:
            check(load().tlab_dns_set_anelastic(self._h, None, None), "tlab_dns_set_anelastic")
            return
        rb = np.ascontiguousarray(rbackground, dtype=np.float64)
        ri = np.ascontiguousarray(1.0 / rb if ribackground is None else ribackground, dtype=np.float64)
        if rb.shape != (self.ny,) or ri.shape != (self.ny,):
            raise TlabError("anelastic profiles: ny values each")
        check(load().tlab_dns_set_anelastic(self._h, rb.ctypes.data_as(dp), ri.ctypes.data_as(dp)), "tlab_dns_set_anelastic")

    def set_fusion(self, on):
        """on (default): pointwise sums folded into the operator kernels; off: the reference's literal sequence."""
        check(load().tlab_dns_set_fusion(self._h, int(bool(on))), "tlab_dns_set_fusion")

    def _arrays(self):
        if self._ptrs is None:
            def arr(ts):
                a = (c_vp * max(len(ts), 1))()
                for i, t in enumerate(ts):
                    a[i] = t.data_ptr()
                return a
            self._ptrs = tuple(arr(t) for t in (self.q, self.s, self.hq, self.hs, self.txc))
        return self._ptrs

    def RHS_GLOBAL_INCOMPRESSIBLE_1(self, dte):
        _use_torch_stream()
        q, s, hq, hs, txc = self._arrays()
        check(load().tlab_rhs_global_incompressible_1(self._h, float(dte), q, s, hq, hs, txc), "tlab_rhs_global_incompressible_1")

    def TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(self, dte, kco=1.0, scale_tendencies=False):
        _use_torch_stream()
        q, s, hq, hs, txc = self._arrays()
        check(load().tlab_time_substep_incompressible_explicit(self._h, float(dte), float(kco), int(scale_tendencies), q, s, hq, hs, txc),
              "tlab_time_substep_incompressible_explicit")

    def load_fields(self, flow_name=None, scal_name=None):
        """Restart files of the reference: <flow_name>.1..3 = u, v, w; <scal_name>.1..nscal (IO_Read_Fields, io_fields.f90:150).
        Returns (nt, params) of the last header read."""
        import torch
        from . import io as tio
        nt, params = None, None
        for name, dst in ((flow_name, self.q), (scal_name, self.s)):
            if name is None or not dst:
                continue
            fields, nt, params = tio.io_read_fields(name, self.nx, self.ny, self.nz, len(dst))
            for t, a in zip(dst, fields):
                t.copy_(torch.from_numpy(a))
        return nt, params

    def save_fields(self, flow_name=None, scal_name=None, nt=0, params=()):
        """IO_Write_Fields (io_fields.f90:346): files the reference's own tools (averages.x, visuals.x, dns.x) read back."""
        from . import io as tio
        for name, src in ((flow_name, self.q), (scal_name, self.s)):
            if name is not None and src:
                tio.io_write_fields(name, self.nx, self.ny, self.nz, nt, [t.cpu().numpy() for t in src], params)

    def TIME_COURANT(self, cfla, cfld):
        """tools/dns/time.f90:365-548.  Returns ((pmax1, pmax2), dtime): the CFL and diffusion maxima and the time step they allow."""
        _use_torch_stream()
        q = self._arrays()[0]
        pmax = (ctypes.c_double * 2)()
        dt = ctypes.c_double(0.0)
        check(load().tlab_time_courant(self._h, q, float(cfla), float(cfld), pmax, ctypes.byref(dt)), "tlab_time_courant")
        return (pmax[0], pmax[1]), dt.value

    def FI_INVARIANT_P(self, result, tmp1):
        """mappings/fi_vectorcalculus.f90:111: result = -div(q)."""
        _use_torch_stream()
        check(load().tlab_fi_invariant_p(self._h, self.q[0].data_ptr(), self.q[1].data_ptr(), self.q[2].data_ptr(), result.data_ptr(), tmp1.data_ptr()),
              "tlab_fi_invariant_p")

    def dilatation_bounds(self):
        """DNS_BOUNDS_CONTROL, tools/dns/dns_local.f90:157-187 (incompressible): (DilMin, DilMax) = logs_data(10:11) of dns.out."""
        self.FI_INVARIANT_P(self.txc[0], self.txc[1])
        mn, mx = ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(load().tlab_minmax(self._h, self.txc[0].data_ptr(), self.nx, self.ny, self.nz, ctypes.byref(mn), ctypes.byref(mx)), "tlab_minmax")
        return -mx.value, -mn.value

    def begin_step(self):
        """hq = hs = 0 of TIME_RUNGEKUTTA (time.f90:212-216) without touching the arrays: the next substep overwrites them."""
        check(load().tlab_dns_begin_step(self._h), "tlab_dns_begin_step")

    def place_arrays(self, pool=40, random_trials=16, dtime=1e-3, seed=0):
        """tlab_dns_place_arrays: q, s, hq, hs, txc move to the allocations (out of a pool of `pool` fresh ones of the txc size) on which the substep runs
        fastest; the fields keep their values, the tendencies are zeroed.  Returns {"ms_first", "ms_best", "ms_median", "ms_worst", "trials", "pool",
        "seconds"}: ms_first / ms_best from the repeats at the end of the search (the allocator's order and the assignment kept, three steps each, back
        to back), median / worst over the single timings of the search."""
        import time
        import torch
        nroles = 2 * (3 + self.nscal) + 9
        m = self.isize_txc_field
        t0 = time.perf_counter()
        state = [t.clone() for t in self.q + self.s]
        dev = self.q[0].device
        self.q = self.s = self.hq = self.hs = self.txc = self._ptrs = None       # the driver's own arrays go back to the allocator before the pool is made
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info()
        pool = max(nroles, min(int(pool), int(0.7 * free / (8.0 * m))))
        cand = []
        try:
            for _ in range(pool):
                cand.append(torch.zeros(m, dtype=torch.float64, device=dev))
        except RuntimeError:      # out of memory: search among what there is (at least the roles themselves must fit, as they did before)
            for _ in range(min(4, max(0, len(cand) - nroles))):      # some headroom for what else the run allocates
                cand.pop()
            torch.cuda.empty_cache()
            if len(cand) < nroles:
                # not even the roles fit beside the saved state: the driver gets arrays of its original sizes back, with its fields, before the error leaves
                cand = None
                torch.cuda.empty_cache()
                ns = self.nscal
                self.q = [r for r in state[:3]]
                self.s = [r for r in state[3:3 + ns]]
                self.hq = [torch.zeros(self.n, dtype=torch.float64, device=dev) for _ in range(3)]
                self.hs = [torch.zeros(self.n, dtype=torch.float64, device=dev) for _ in range(ns)]
                self.txc = [torch.zeros(m, dtype=torch.float64, device=dev) for _ in range(9)]
                self._ptrs = None
                raise
            pool = len(cand)
        parr = (c_vp * pool)(*[t.data_ptr() for t in cand])
        sarr = (c_vp * len(state))(*[t.data_ptr() for t in state])
        assign = (ctypes.c_int * nroles)()
        rep = (ctypes.c_double * 5)()
        _use_torch_stream()
        rc = load().tlab_dns_place_arrays(self._h, pool, parr, sarr, float(dtime), int(random_trials), int(seed), assign, rep)
        if rc != 0:      # the driver stays usable: the pool in order
            for i in range(nroles):
                assign[i] = i
        a = [cand[i] for i in assign]
        ns = self.nscal
        self.q, self.s = [t[: self.n] for t in a[0:3]], [t[: self.n] for t in a[3:3 + ns]]
        self.hq, self.hs = [t[: self.n] for t in a[3 + ns:6 + ns]], [t[: self.n] for t in a[6 + ns:6 + 2 * ns]]
        self.txc = a[6 + 2 * ns:]
        self._ptrs = None
        for t, r in zip(self.q + self.s, state):
            t.copy_(r)
        for t in self.hq + self.hs:
            t.zero_()
        del cand, state, a
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        check(rc, "tlab_dns_place_arrays")
        return {"ms_first": rep[0], "ms_best": rep[1], "ms_median": rep[2], "ms_worst": rep[3], "trials": int(rep[4]), "pool": pool,
                "seconds": time.perf_counter() - t0}

    def TIME_RUNGEKUTTA(self, dtime):
        """One time step: hq = hs = 0, then rkm_endstep substeps (time.f90:212-298)."""
        self.begin_step()
        for k in range(self.rkm_endstep):
            last = k == self.rkm_endstep - 1
            self.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * self.kdt[k], 1.0 if last else self.kco[k], not last)

    def __del__(self):
        try:
            if self._h:
                load().tlab_dns_destroy(self._h)
        except Exception:
            pass
