"""ctypes front end of oracle/tlab_cpu.c, the C + OpenMP restatement of the reference's CPU algorithm for the hot path.

TEST INFRASTRUCTURE / CPU BASELINE ONLY (tests/ and bench.py's cpu_baseline leg; never imported by tlab_amd/).
Plans (coefficient tables, per-mode integral systems) are built by the numpy oracle -- FDM_CreatePlan / FDM_Int1_Initialize are
start-up work in the reference too -- and handed to the C code, which runs the per-substep path on all host cores.
Build: `make -C oracle cpu` (gcc -O3 -fopenmp, no FMA contraction) -> oracle/libtlab_cpu.so.
"""
import ctypes
import os

import numpy as np

from . import tlab_oracle as O
from . import tlab_oracle_poisson as OP

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
c_int, c_dbl, c_vp, c_ll = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_longlong


class CpuFdm(ctypes.Structure):
    _fields_ = [("n", c_int), ("periodic", c_int), ("need_1der", c_int), ("rhs1", c_vp), ("rhs_b1", c_vp), ("rhs_t1", c_vp), ("lu1", c_vp),
                ("rhs2", c_vp), ("lu2", c_vp)]


class CpuDns(ctypes.Structure):
    _fields_ = [("nx", c_int), ("ny", c_int), ("nz", c_int), ("nscal", c_int), ("g", c_vp * 3), ("lu2d", (c_vp * 8) * 3), ("poisson", c_vp)]


def lib_path():
    return os.path.join(_HERE, "libtlab_cpu.so")


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(lib_path()):
            raise RuntimeError("%s not built: make -C oracle cpu" % lib_path())
        L = ctypes.CDLL(lib_path())
        L.tlabcpu_poisson_create.restype = c_vp
        L.tlabcpu_poisson_create.argtypes = [c_int, c_int, c_int, c_ll] + [c_vp] * 8 + [c_ll] + [c_vp] * 9
        L.tlabcpu_poisson_destroy.argtypes = [c_vp]
        L.tlabcpu_opr_partial.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.tlabcpu_opr_burgers.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.tlabcpu_opr_poisson.argtypes = [c_vp] * 8
        L.tlabcpu_time_substep.argtypes = [c_vp, c_dbl, c_dbl, c_int] + [c_vp] * 9
        L.tlabcpu_transpose.argtypes = [c_vp, c_int, c_int, c_vp]
        L.tlabcpu_num_threads.restype = c_int
        L.tlabcpu_set_num_threads.argtypes = [c_int]
        L.tlabcpu_fill.argtypes = [c_vp, c_ll, c_dbl]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_vp) if a is not None else None


class FdmTables:
    """The tables of one oracle FdmPlan in the layouts tlab_cpu.c reads (column-major lhs/rhs/lu like the Fortran arrays)."""

    def __init__(self, g):
        d1, d2 = g.der1, g.der2
        if d1.nb_diag != (3, 5) or d2.nb_diag != (3, 7) or getattr(d2, "direct", False):
            raise NotImplementedError("tlab_cpu.c restates the default schemes (CompactJacobian6 / CompactJacobian6Hyper) only")
        self.g = g
        self.keep = [np.asfortranarray(d1.rhs[:, :5]), np.ascontiguousarray(d1.rhs_b, dtype=np.float64), np.ascontiguousarray(d1.rhs_t, dtype=np.float64),
                     np.asfortranarray(d1.lu), np.asfortranarray(d2.rhs[:, :10]), np.asfortranarray(d2.lu)]
        assert self.keep[1].shape == (4, 8) and self.keep[2].shape == (5, 7)
        self.c = CpuFdm(g.size, int(g.periodic), int(d2.need_1der), *[_p(a) for a in self.keep])
        self._lu2d = {}

    def diffusion_lu(self, nu):
        """fdmDiffusion(ig)%lu(:,:,is), physics/opr_burgers.f90:100-111"""
        if nu not in self._lu2d:
            self._lu2d[nu] = np.asfortranarray(self.g.diffusion_lu(nu))
        return self._lu2d[nu]


class CpuPoisson:
    """OPR_Elliptic_Initialize (operators/opr_elliptic.f90:86-250): per-mode first-order integral systems, factorized once."""

    def __init__(self, gx, gy, gz, nx, ny, nz, chunk=8192):
        plan = OP.PoissonPlan(gx, gy, gz, nx, ny, nz)
        nxh = plan.nxh
        M = nz * nxh
        lam = np.sqrt(plan.lam2.reshape(M))
        sing = plan.sing.reshape(M)
        self.nx, self.ny, self.nz = nx, ny, nz
        self.isize_txc_field = (nx + 2) * ny * nz
        sets = []
        for sel in (~sing, sing):
            idx = np.nonzero(sel)[0].astype(np.int32)
            n = idx.shape[0]
            lhs = [np.empty((n, 5, ny)), np.empty((n, 5, ny))]
            rb = [np.empty((n, 3, 4)), np.empty((n, 3, 4))]
            rt = [np.empty((n, 3, 4)), np.empty((n, 3, 4))]
            rhs = [None, None]
            for a in range(0, n, chunk):                     # the oracle vectorises over modes; chunks bound its temporaries
                l = lam[idx[a:a + chunk]]
                for b, (sign, bc) in enumerate(((1.0, O.BCS_MIN), (-1.0, O.BCS_MAX))):
                    p = OP.int1_initialize(gy.der1, sign * l, bc)
                    lhs[b][a:a + chunk] = p.lhs.transpose(2, 1, 0)
                    rb[b][a:a + chunk] = p.rhs_b[0:3, 0:4, :].transpose(2, 0, 1)
                    rt[b][a:a + chunk] = p.rhs_t[0:3, 0:4, :].transpose(2, 0, 1)
                    rhs[b] = np.ascontiguousarray(p.rhs[:, :3].T)
            if n == 0:
                for b, bc in enumerate((O.BCS_MIN, O.BCS_MAX)):
                    rhs[b] = np.ascontiguousarray(OP.int1_create_system(gy.der1, np.zeros(1), bc).rhs[:, :3].T)
            sets.append((idx, lhs, rb, rt, rhs))
        (ri, rl, rrb, rrt, rhs), (si, sl, srb, srt, rhs_s) = sets
        if rhs[0] is None:
            rhs = rhs_s
        self.lam = np.ascontiguousarray(lam[ri])
        self.keep = [ri, self.lam, rl, rrb, rrt, si, sl, srb, srt, rhs]
        L = load()
        self.h = L.tlabcpu_poisson_create(nx, ny, nz, ri.shape[0], _p(ri), _p(self.lam), _p(rl[0]), _p(rl[1]), _p(rrb[0]), _p(rrb[1]), _p(rrt[0]), _p(rrt[1]),
                                          si.shape[0], _p(si), _p(sl[0]), _p(sl[1]), _p(srb[0]), _p(srb[1]), _p(srt[0]), _p(srt[1]), _p(rhs[0]), _p(rhs[1]))

    def __del__(self):
        try:
            load().tlabcpu_poisson_destroy(self.h)
        except Exception:
            pass


def opr_partial(idir, itype, nx, ny, nz, ibc, t, u):
    """OPR_Partial_X/Y/Z on the C path.  t: FdmTables.  Returns (result, tmp1)."""
    n = nx * ny * nz
    u = np.ascontiguousarray(u, dtype=np.float64)
    res, tmp1, wrk3d, wrk2d = np.empty(n), np.empty(n), np.empty(n), np.empty(n // (nx, ny, nz)[idir - 1])
    rc = load().tlabcpu_opr_partial(idir, itype, nx, ny, nz, ibc, ctypes.addressof(t.c), _p(u), _p(res), _p(tmp1), _p(wrk3d), _p(wrk2d))
    assert rc == 0, rc
    return res, (tmp1 if itype == O.OPR_P2_P1 else None)


def opr_burgers(idir, nx, ny, nz, ibc, t, nu, s, vel):
    """OPR_Burgers_X/Y/Z (OPR_B_U_IN with the transposed velocity built here the way an earlier OPR_B_SELF call leaves it)."""
    n = nx * ny * nz
    s = np.ascontiguousarray(s, dtype=np.float64)
    vel = np.ascontiguousarray(vel, dtype=np.float64)
    L = load()
    vt = np.empty(n)
    if idir == 1:
        L.tlabcpu_transpose(_p(vel), nx, ny * nz, _p(vt))
    elif idir == 2:
        L.tlabcpu_transpose(_p(vel), nx * ny, nz, _p(vt))
    else:
        vt = vel
    res, tmp1, wrk3d, wrk2d = np.empty(n), np.empty(n), np.empty(n), np.empty(n // (nx, ny, nz)[idir - 1])
    lu = t.diffusion_lu(nu)
    rc = L.tlabcpu_opr_burgers(idir, 1, nx, ny, nz, ibc, ctypes.addressof(t.c), _p(lu), _p(s), _p(res), _p(tmp1), _p(vt), _p(wrk3d), _p(wrk2d))
    assert rc == 0, rc
    return res


def opr_poisson(P, f, bcs_hb, bcs_ht):
    """OPR_Poisson (BCS_NN).  Returns (p, dpdy)."""
    n = P.nx * P.ny * P.nz
    p = np.array(f, dtype=np.float64, copy=True)
    t1, t2, w3 = (np.empty(P.isize_txc_field) for _ in range(3))
    dpdy = np.empty(n)
    hb, ht = np.ascontiguousarray(bcs_hb, dtype=np.float64), np.ascontiguousarray(bcs_ht, dtype=np.float64)
    rc = load().tlabcpu_opr_poisson(P.h, _p(p), _p(t1), _p(t2), _p(w3), _p(hb), _p(ht), _p(dpdy))
    assert rc == 0
    return p, dpdy


class CpuDnsDriver:
    """q, s, hq, hs, txc + RHS_GLOBAL_INCOMPRESSIBLE_1 / RK update on the C path (no-slip walls, Dirichlet scalars)."""

    def __init__(self, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=None):
        assert nscal <= 7
        self.nx, self.ny, self.nz = len(x), len(y), len(z)
        self.n = self.nx * self.ny * self.nz
        self.nscal = nscal
        h = hyper_bc1_ext
        self.g = [O.FdmPlan(x, True, True, hyper_bc1_ext=h), O.FdmPlan(y, False, yuniform, hyper_bc1_ext=h), O.FdmPlan(z, True, True, hyper_bc1_ext=h)]
        self.t = [FdmTables(g) for g in self.g]
        self.poisson = CpuPoisson(self.g[0], self.g[1], self.g[2], self.nx, self.ny, self.nz)
        nt = self.poisson.isize_txc_field
        def zeros(m):                                   # first touch in parallel (NUMA placement), tlab_cpu.c tlabcpu_fill
            a = np.empty(m)
            load().tlabcpu_fill(_p(a), m, 0.0)
            return a
        self.q = [zeros(self.n) for _ in range(3)]
        self.s = [zeros(self.n) for _ in range(nscal)]
        self.hq = [zeros(self.n) for _ in range(3)]
        self.hs = [zeros(self.n) for _ in range(nscal)]
        self.txc = [zeros(nt) for _ in range(9)]
        self.wrk3d = zeros(nt)
        self.wrk2d = np.zeros(max(self.n // self.nx, self.n // self.ny, self.n // max(self.nz, 1)))
        self.hb, self.ht = np.zeros(self.nx * self.nz), np.zeros(self.nx * self.nz)
        self.c = CpuDns()
        self.c.nx, self.c.ny, self.c.nz, self.c.nscal = self.nx, self.ny, self.nz, nscal
        nus = [visc] + [visc / schmidt[i] for i in range(nscal)]
        self._lus = []
        for d in range(3):
            self.c.g[d] = ctypes.addressof(self.t[d].c)
            for i, nu in enumerate(nus):
                lu = self.t[d].diffusion_lu(nu)
                self._lus.append(lu)
                self.c.lu2d[d][i] = lu.ctypes.data
        self.c.poisson = self.poisson.h

    def _arr(self, lst):
        a = (c_vp * max(len(lst), 1))()
        for i, t in enumerate(lst):
            a[i] = t.ctypes.data
        return a

    def time_substep(self, dte, kco=1.0, scale=False):
        rc = load().tlabcpu_time_substep(ctypes.addressof(self.c), float(dte), float(kco), int(scale), self._arr(self.q), self._arr(self.s),
                                         self._arr(self.hq), self._arr(self.hs), self._arr(self.txc), _p(self.wrk3d), _p(self.wrk2d), _p(self.hb), _p(self.ht))
        assert rc == 0


def host_description():
    """CPU model and core count of this host (for bench.py's cpu_baseline record)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, (os.cpu_count() or 1)
