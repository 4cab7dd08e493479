"""Host-side mirror of the NATIVE z-slab driver (tlab_amd/csrc/slab.cpp behind tlab_slab_dns_* of include/tlab_amd.h): the decomposed
RHS_GLOBAL_INCOMPRESSIBLE_1 / TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT a Fortran / MPI host calls when ims_npro_k > 1, and what `bench.py --gpus N` runs.

Python only allocates the module arrays (plain torch tensors: the neighbours' halo planes live in buffers of the driver), picks the transport and
forwards the calls; the operator sequence, the exchanges and their overlap are in C++.  Transports (tlab_slab_transport, include/tlab_amd.h):

    "loopback"  all P ranks inside this process on one device, exchanges = device copies (verification on one GPU; tests/test_gpu_slab.py)
    "rccl"      one rank per process, grouped ncclSend / ncclRecv on the communication stream of libtlab_amd_comm.so (the product path)
    "dist"      one rank per process, the five entry points as ctypes callbacks over torch.distributed with host-staged payloads (gloo): several
                processes sharing ONE GPU, where RCCL refuses to run -- a functional test of the per-process code path, and the shape of what a
                Fortran host with a GPU-aware MPI would hand in.

`tlab_amd/parallel.py::SlabDns` (Python over torch.distributed) remains as the cross-check of this driver and for thin slabs, which need the
reference's K-transposition scheme."""
import ctypes
import numpy as np

from .lib import load, check, TlabError, c_vp, c_int
from .operators import FdmPlan, _use_torch_stream
from .dns import rk_coefficients, RKM_EXP3, DNS_BCS_DIRICHLET, _bcs_arrays

_pp = ctypes.POINTER(c_vp)
_pll = ctypes.POINTER(ctypes.c_longlong)
RING_FN = ctypes.CFUNCTYPE(c_int, c_vp, c_vp, c_int, _pll, _pp, _pp, _pp, _pp)
A2A_FN = ctypes.CFUNCTYPE(c_int, c_vp, c_vp, _pp, _pll, _pp, _pll)
WAIT_FN = ctypes.CFUNCTYPE(c_int, c_vp, c_vp, c_int)
RED_FN = ctypes.CFUNCTYPE(c_int, c_vp, ctypes.POINTER(ctypes.c_double), c_int, c_int)
DESTROY_FN = ctypes.CFUNCTYPE(None, c_vp)


class SlabTransport(ctypes.Structure):
    """struct tlab_slab_transport"""
    _fields_ = [("ctx", c_vp), ("nranks", c_int), ("nlocal", c_int), ("first", c_int), ("ring_start", RING_FN), ("alltoallv_start", A2A_FN),
                ("wait", WAIT_FN), ("allreduce", RED_FN), ("destroy", DESTROY_FN)]


def loopback_transport(P):
    t = SlabTransport()
    check(load().tlab_slab_transport_loopback(ctypes.byref(t), int(P)), "tlab_slab_transport_loopback")
    return t, None


def rccl_transport(group=None):
    """The z communicator over RCCL inside libtlab_amd_comm.so; the ncclUniqueId travels through torch.distributed's object broadcast (a Fortran host
    uses MPI_Bcast).  Returns (transport, keep-alive)."""
    import torch.distributed as dist
    from . import comm as C
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [C.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    nc = C.NativeComm(box[0], world, rank, 1, world)
    t = SlabTransport()
    C.check(C.load().tlab_comm_slab_transport(nc._h, ctypes.byref(t)), "tlab_comm_slab_transport")
    return t, nc


def dist_transport(group=None):
    """The five entry points as callbacks over torch.distributed with host-staged payloads.  Synchronous: every exchange is complete when its
    start returns (ticket 0), so nothing overlaps -- functional runs only."""
    import torch
    import torch.distributed as dist
    L = load()
    rank, P = dist.get_rank(group), dist.get_world_size(group)

    def d2h(ptr, count):
        h = torch.empty(int(count), dtype=torch.float64)
        check(L.tlab_memcpy_d2h(c_vp(h.data_ptr()), c_vp(ptr), int(count) * 8), "tlab_memcpy_d2h")
        return h

    def h2d(ptr, h):
        check(L.tlab_memcpy_h2d(c_vp(ptr), c_vp(h.data_ptr()), h.numel() * 8), "tlab_memcpy_h2d")

    def ring(ctx, stream, nmsg, count, to_left, to_right, from_right, from_left):
        try:
            left, right = (rank - 1) % P, (rank + 1) % P
            L.tlab_sync()
            sl = [d2h(to_left[i], count[i]) for i in range(nmsg)]
            sr = [d2h(to_right[i], count[i]) for i in range(nmsg)]
            rr = [torch.empty(int(count[i]), dtype=torch.float64) for i in range(nmsg)]
            rl = [torch.empty(int(count[i]), dtype=torch.float64) for i in range(nmsg)]
            ops = [dist.P2POp(dist.isend, t, left, group) for t in sl] + [dist.P2POp(dist.isend, t, right, group) for t in sr]
            ops += [dist.P2POp(dist.irecv, t, right, group) for t in rr] + [dist.P2POp(dist.irecv, t, left, group) for t in rl]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for i in range(nmsg):
                h2d(from_right[i], rr[i])
                h2d(from_left[i], rl[i])
            return 0
        except Exception as e:       # noqa: BLE001  (a callback must not raise through C)
            print("dist_transport ring:", e, flush=True)
            return -3

    def a2a(ctx, stream, send, scount, recv, rcount):
        try:
            L.tlab_sync()
            sc, rc = [int(scount[p]) for p in range(P)], [int(rcount[p]) for p in range(P)]
            hs = d2h(send[0], sum(sc))
            hr = torch.empty(sum(rc), dtype=torch.float64)
            dist.all_to_all_single(hr, hs, output_split_sizes=rc, input_split_sizes=sc, group=group)
            h2d(recv[0], hr)
            return 0
        except Exception as e:       # noqa: BLE001
            print("dist_transport alltoallv:", e, flush=True)
            return -3

    def wait(ctx, stream, ticket):
        return 0

    def red(ctx, values, n, op):
        try:
            t = torch.tensor([values[i] for i in range(n)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 0 else (dist.ReduceOp.MIN if op == 1 else dist.ReduceOp.SUM), group=group)
            for i in range(n):
                values[i] = float(t[i])
            return 0
        except Exception as e:       # noqa: BLE001
            print("dist_transport allreduce:", e, flush=True)
            return -3

    fns = (RING_FN(ring), A2A_FN(a2a), WAIT_FN(wait), RED_FN(red))
    t = SlabTransport(None, P, 1, rank, fns[0], fns[1], fns[2], fns[3], DESTROY_FN())
    return t, fns


class NativeSlabDns:
    """tlab_slab_dns_* with the interface of tlab_amd.parallel.SlabDns (st[rank][name][i] tensors, scatter, substep_of_cycle, monitors)."""

    zmode = "halo"

    def __init__(self, transport, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3, hyper_bc1_ext=0.0,
                 device="cuda", plans=None, gy_elliptic=None, size=None, group=None, fused_x=None):
        """transport: "loopback" (size = number of ranks), "rccl" or "dist" (torch.distributed initialised; one rank per process).
        fused_x: None = the library's default (TLAB_SLAB_FUSED_X); False keeps the separate repack passes and rocFFT's inverse x-transform,
        whose results equal tlab_amd.parallel.SlabDns to the bit."""
        import os
        import torch
        L = load()
        if transport == "loopback":
            self._tr, self._keep = loopback_transport(size)
        elif transport == "rccl":
            self._tr, self._keep = rccl_transport(group)
        elif transport == "dist":
            self._tr, self._keep = dist_transport(group)
        else:
            raise TlabError("transport: loopback, rccl or dist")
        self.transport = transport
        P = self._tr.nranks
        self.size = P
        self.local_ranks = list(range(self._tr.first, self._tr.first + self._tr.nlocal))
        self.nx, self.ny, self.nzt = len(x), len(y), len(z)
        self.nscal, self.visc = int(nscal), float(visc)
        self.schmidt = [float(v) for v in schmidt][: self.nscal]
        self.g = list(plans) if plans is not None else [
            FdmPlan(x, True, True, hyper_bc1_ext=hyper_bc1_ext), FdmPlan(y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext),
            FdmPlan(z, True, True, hyper_bc1_ext=hyper_bc1_ext)]
        self.gy_elliptic = gy_elliptic
        self.kdt, self.kco = rk_coefficients(rkm_mode)
        self.rkm_endstep = len(self.kdt)
        sc = np.ascontiguousarray(self.schmidt if self.nscal else [1.0], dtype=np.float64)
        self._h = c_vp(0)
        saved = os.environ.get("TLAB_SLAB_FUSED_X")
        if fused_x is not None:
            os.environ["TLAB_SLAB_FUSED_X"] = "1" if fused_x else "0"
        try:
            rc = L.tlab_slab_dns_create(ctypes.byref(self._h), ctypes.byref(self._tr), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                        self.nzt, self.nscal, self.visc, sc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                        gy_elliptic._h if gy_elliptic is not None else None)
        finally:
            if fused_x is not None:
                if saved is None:
                    os.environ.pop("TLAB_SLAB_FUSED_X", None)
                else:
                    os.environ["TLAB_SLAB_FUSED_X"] = saved
        if rc != 0:
            # a refused configuration leaves the transport context with the caller (include/tlab_amd.h): release it before raising
            try:
                if self._tr.destroy:
                    self._tr.destroy(self._tr.ctx)
            finally:
                if self.transport == "rccl" and self._keep is not None:
                    self._keep.close()
                self._keep = None
                self._h = c_vp(0)
        check(rc, "tlab_slab_dns_create")
        self.fused_x = bool(L.tlab_slab_dns_info(self._h, 6))
        self.kmax = int(L.tlab_slab_dns_info(self._h, 0))
        self.stages = int(L.tlab_slab_dns_info(self._h, 4))
        self.npage = self.nx * self.ny
        self.n = self.npage * self.kmax
        self.isize_txc = (self.nx + 2) * self.ny * self.kmax
        self.st = {}
        for l, r in enumerate(self.local_ranks):
            def field(m):
                return torch.zeros(m, dtype=torch.float64, device=device)
            S = {name: [field(m) for _ in range(cnt)] for name, cnt, m in
                 (("q", 3, self.n), ("s", self.nscal, self.n), ("hq", 3, self.n), ("hs", self.nscal, self.n), ("txc", 9, self.isize_txc))}
            arr = lambda ts: (c_vp * max(len(ts), 1))(*[t.data_ptr() for t in ts])       # noqa: E731
            check(L.tlab_slab_dns_bind(self._h, l, arr(S["q"]), arr(S["s"]), arr(S["hq"]), arr(S["hs"]), arr(S["txc"])), "tlab_slab_dns_bind")
            self.st[r] = S

    def redraw_arrays(self, pool=34, seed=0):
        """Every local rank's q, s, hq, hs, txc move to allocations drawn at random from a pool of `pool` fresh ones per rank; the fields keep their
        values (tlab_amd/placement.py; what bench.py does before its timed region on slabs)."""
        from .placement import redraw_rank_arrays
        return redraw_rank_arrays(self, "tlab_slab_dns_bind", pool=pool, seed=seed)

    def set_bcs(self, velocity_jmin="noslip", velocity_jmax="noslip", scalar_jmin="dirichlet", scalar_jmax="dirichlet"):
        """As Dns.set_bcs (dns.ini [BoundaryConditions] keywords, boundary_bcs.f90:102-190)."""
        fj0, fj1, sj0, sj1 = _bcs_arrays(self.nscal, velocity_jmin, velocity_jmax, scalar_jmin, scalar_jmax)
        ia = lambda v, m: (c_int * max(m, 1))(*list(v)[:m])       # noqa: E731
        check(load().tlab_slab_dns_set_bcs(self._h, ia(fj0, 3), ia(fj1, 3), ia(sj0, self.nscal), ia(sj1, self.nscal)), "tlab_slab_dns_set_bcs")

    def begin_step(self):
        check(load().tlab_slab_dns_begin_step(self._h), "tlab_slab_dns_begin_step")

    def set_surface_bcs(self, sfc_jmin=None, sfc_jmax=None, coupling_jmin=None, coupling_jmax=None):
        """As Dns.set_surface_bcs: Scalar<i>SfcTypeJmin/Jmax = "static" | "linear", Scalar<i>CouplingJmin/Jmax (boundary_bcs.f90:76-87)."""
        ns = max(self.nscal, 1)
        code = lambda v: [1 if str(t).lower() == "linear" else 0 for t in (v or ["static"] * ns)]      # noqa: E731
        s0, s1 = (c_int * ns)(*code(sfc_jmin)[:ns]), (c_int * ns)(*code(sfc_jmax)[:ns])
        c0 = (ctypes.c_double * ns)(*[float(v) for v in (coupling_jmin or [0.0] * ns)][:ns])
        c1 = (ctypes.c_double * ns)(*[float(v) for v in (coupling_jmax or [0.0] * ns)][:ns])
        check(load().tlab_slab_dns_set_surface_bcs(self._h, s0, s1, c0, c1), "tlab_slab_dns_set_surface_bcs")

    def set_remove_divergence(self, on):
        """dns.ini [Main] TermDivergence (as Dns.set_remove_divergence): off = the forcing of the pressure equation is div(hq) alone."""
        check(load().tlab_slab_dns_set_remove_divergence(self._h, int(bool(on))), "tlab_slab_dns_set_remove_divergence")

    def RHS_GLOBAL_INCOMPRESSIBLE_1(self, dte):
        _use_torch_stream()
        check(load().tlab_slab_dns_rhs(self._h, float(dte)), "tlab_slab_dns_rhs")

    def TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(self, dte, kco=1.0, scale_tendencies=False):
        _use_torch_stream()
        check(load().tlab_slab_dns_substep(self._h, float(dte), float(kco), int(scale_tendencies)), "tlab_slab_dns_substep")

    def substep_of_cycle(self, k, dtime):
        """k-th substep of consecutive RK steps (the tendencies count as zero at the start of each step, time.f90:212-216)."""
        s = k % self.rkm_endstep
        if s == 0:
            self.begin_step()
        last = s == self.rkm_endstep - 1
        self.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * self.kdt[s], 1.0 if last else self.kco[s], not last)

    def TIME_COURANT(self, cfla, cfld):
        _use_torch_stream()
        pmax, dt = (ctypes.c_double * 2)(), ctypes.c_double(0.0)
        check(load().tlab_slab_dns_time_courant(self._h, float(cfla), float(cfld), pmax, ctypes.byref(dt)), "tlab_slab_dns_time_courant")
        return (pmax[0], pmax[1]), dt.value

    def dilatation_bounds(self):
        _use_torch_stream()
        mn, mx = ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(load().tlab_slab_dns_dilatation_bounds(self._h, ctypes.byref(mn), ctypes.byref(mx)), "tlab_slab_dns_dilatation_bounds")
        return mn.value, mx.value

    def scatter(self, name, idx, global_field):
        """global_field: flat tensor nx*ny*nz_total (x fastest); every local rank takes its planes."""
        for r in self.local_ranks:
            self.st[r][name][idx].copy_(global_field[r * self.n:(r + 1) * self.n])

    def close(self):
        if self._h:
            load().tlab_slab_dns_destroy(self._h)
            self._h = c_vp(0)
        if self.transport == "rccl" and self._keep is not None:
            self._keep.close()
            self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass
