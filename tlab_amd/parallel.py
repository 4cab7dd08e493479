"""z-slab domain decomposition of the hot path over the GPUs of one node.

Mirrors the reference's MPI layer for the 1 x npro_k decomposition (ims_npro_i = 1): each rank (= one process = one GPU) owns
kmax = nz/npro_k planes of every field; x- and y-operators are local; z-operators and the z-FFT of the Poisson solver work on
the K-transposed layout obtained with one all-to-all inside the z communicator:

    TLabMPI_Trp_ExecK_Forward / _Backward     base/tlab_mpi_transpose.f90:343-458, plan :290-339
      local a(npage, kmax), npage = imax*jmax  ->  b(nlines, kmax*npro_k), nlines = npage/npro_k
      peer p receives the in-plane block [p*nlines, (p+1)*nlines) of all my planes and stores it at z-offset my_rank*kmax.

The collective is torch.distributed.all_to_all_single (RCCL over xGMI on the GPU box, gloo in the CPU tests); the pack/unpack
around it are strided torch copies (pure index work, bit-exact).  All arithmetic stays in the HIP library (C ABI): in the
transposed layout a z-operator is just the dir = 3 operator on a (nlines, 1, nz_total) box.

`LoopbackComm` runs all npro_k ranks inside ONE process on one device, exchanging blocks by direct copies: the complete
decomposed algorithm (index maps, per-rank wavenumber offsets, singular-mode ownership) can then be verified against the
single-domain result on a single GPU (tests/test_gpu_slab.py) although only the driver's 8-GPU node can run it for real.
"""
import ctypes
import numpy as np

from .lib import load, check, TlabError, c_vp
from .operators import FdmPlan, _use_torch_stream, _ptr
from .dns import rk_coefficients, RKM_EXP3, DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN, _bcs_arrays


# ------------------------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------------------------
class DistComm:
    """ims_comm_z over torch.distributed: one rank per process."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]

    def all_to_all(self, sends):
        """sends: {rank: tensor [size, chunk]} (contiguous).  Returns {rank: tensor [size(src), chunk]}."""
        import torch
        (r, s), = sends.items()
        out = torch.empty_like(s)
        self.dist.all_to_all_single(out, s, group=self.group)
        return {r: out}


class LoopbackComm:
    """All ranks of the z communicator simulated in this process (verification only)."""

    def __init__(self, size):
        self.size = int(size)
        self.rank = 0
        self.local_ranks = list(range(self.size))

    def all_to_all(self, sends):
        import torch
        out = {}
        for r in self.local_ranks:
            out[r] = torch.stack([sends[p][r] for p in range(self.size)])
        return out


# ------------------------------------------------------------------------------------------------------------------
# K-transposes (pure index work)
# ------------------------------------------------------------------------------------------------------------------
def trp_k_forward(comm, a, npage, kmax, width=1):
    """TLabMPI_Trp_ExecK_Forward (tlab_mpi_transpose.f90:343-399).  a: {rank: flat tensor of npage*kmax*width}; width = 2 for
    complex data (MPI_DOUBLE_COMPLEX plans of opr_fourier.f90:86-88).  Returns {rank: flat b(nlines, kmax*size)}."""
    P = comm.size
    if npage % P != 0:
        raise TlabError("K-transposition: npage must be divisible by the number of z slabs (tlab_mpi_transpose.f90:292)")
    nl = npage // P * width
    sends = {r: a[r].view(kmax, P, nl).transpose(0, 1).contiguous() for r in comm.local_ranks}   # [peer][k][line]
    recv = comm.all_to_all({r: s.view(P, kmax * nl) for r, s in sends.items()})
    return {r: recv[r].reshape(-1) for r in comm.local_ranks}     # [src][k][line] == b(line, z = src*kmax + k)


def trp_k_backward(comm, b, npage, kmax, width=1):
    """TLabMPI_Trp_ExecK_Backward (tlab_mpi_transpose.f90:403-458): exact inverse of trp_k_forward."""
    P = comm.size
    nl = npage // P * width
    recv = comm.all_to_all({r: b[r].view(P, kmax * nl) for r in comm.local_ranks})      # block p of b = planes of rank p
    return {r: recv[r].view(P, kmax, nl).transpose(0, 1).reshape(-1) for r in comm.local_ranks}   # a(k, src*nlines + l)


# ------------------------------------------------------------------------------------------------------------------
# the decomposed RK substep
# ------------------------------------------------------------------------------------------------------------------
class SlabDns:
    """RHS_GLOBAL_INCOMPRESSIBLE_1 + TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT with ims_npro_k = comm.size.
    Same operator sequence as tlab_amd/csrc/rhs.cpp (= tools/dns/rhs_global_incompressible_1.f90:98-375); the z-operators
    go through the K-transposes, the transposed w is kept and reused like the reference does with tmp6 (:100,104,116,152)."""

    def __init__(self, comm, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3,
                 hyper_bc1_ext=0.1, device="cuda"):
        import torch
        self.comm = comm
        P = comm.size
        self.nx, self.ny, self.nzt = len(x), len(y), len(z)
        if self.nzt % P:
            raise TlabError("nz must be divisible by the number of z slabs")
        self.kmax = self.nzt // P
        self.npage = self.nx * self.ny
        self.nxh = self.nx // 2 + 1
        self.n = self.npage * self.kmax
        if self.npage % P or (self.nxh * self.ny) % P:
            raise TlabError("imax*jmax and (imax/2+1)*jmax must be divisible by the number of z slabs")
        self.nlines = self.npage // P
        self.nscal, self.visc = int(nscal), float(visc)
        self.schmidt = [float(v) for v in schmidt][: self.nscal]
        self.g = [FdmPlan(x, True, True, hyper_bc1_ext=hyper_bc1_ext), FdmPlan(y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext),
                  FdmPlan(z, True, True, hyper_bc1_ext=hyper_bc1_ext)]
        self.kdt, self.kco = rk_coefficients(rkm_mode)
        self.rkm_endstep = len(self.kdt)
        self.flow_jmin, self.flow_jmax = [DNS_BCS_DIRICHLET] * 3, [DNS_BCS_DIRICHLET] * 3
        self.scal_jmin, self.scal_jmax = [DNS_BCS_DIRICHLET] * self.nscal, [DNS_BCS_DIRICHLET] * self.nscal
        self.isize_txc = (self.nx + 2) * self.ny * self.kmax
        L = load()
        self.st = {}
        for r in comm.local_ranks:
            z0 = lambda m: torch.zeros(m, dtype=torch.float64, device=device)   # noqa: E731
            h = c_vp(0)
            check(L.tlab_poisson_plan_create_slab(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                                  self.kmax, self.nzt, r * self.kmax, P), "tlab_poisson_plan_create_slab")
            self.st[r] = dict(q=[z0(self.n) for _ in range(3)], s=[z0(self.n) for _ in range(self.nscal)],
                              hq=[z0(self.n) for _ in range(3)], hs=[z0(self.n) for _ in range(self.nscal)],
                              txc=[z0(self.isize_txc) for _ in range(9)], rt=z0(self.n), hb=z0(self.nx * self.kmax),
                              ht=z0(self.nx * self.kmax), poisson=h)

    # ---- thin wrappers over the C ABI -------------------------------------------------------------------------------
    def _burgers(self, d, g, ivel, nx, ny, nz, nu, s, u, res, tmp):
        check(load().tlab_opr_burgers(d, g._h, ivel, nx, ny, nz, 0, float(nu), _ptr(s), _ptr(u), _ptr(res), _ptr(tmp), 0), "tlab_opr_burgers")

    def _partial(self, d, g, nx, ny, nz, u, res):
        check(load().tlab_opr_partial(d, g._h, 1, nx, ny, nz, 0, _ptr(u), _ptr(res), c_vp(0)), "tlab_opr_partial")

    def _local(self, fn):
        for r in self.comm.local_ranks:
            fn(r, self.st[r])

    # ---- z-operators through the K-transposes -------------------------------------------------------------------------
    def burgers_z(self, nu, s_key, res_idx, self_vel=False):
        """OPR_Burgers_Z with ims_npro_k > 1 (opr_burgers.f90:386-426).  s_key: callable state -> operand tensor.
        The advecting velocity is always w; its transposed copy w_t is built by the SELF call and reused."""
        c = self.comm
        st = trp_k_forward(c, {r: s_key(self.st[r])[: self.n] for r in c.local_ranks}, self.npage, self.kmax)
        if self_vel:
            for r in c.local_ranks:
                self.st[r]["w_t"] = st[r]
        res = {}
        for r in c.local_ranks:
            S = self.st[r]
            res[r] = S["rt"]
            self._burgers(3, self.g[2], 0 if self_vel else 1, self.nlines, 1, self.nzt, nu, st[r], S["w_t"], res[r], S["txc"][8][: self.n])
        back = trp_k_backward(c, res, self.npage, self.kmax)
        for r in c.local_ranks:
            self.st[r]["txc"][res_idx][: self.n].copy_(back[r])

    def partial_z(self, src_idx, dst_idx):
        """OPR_Partial_Z(OPR_P1) with ims_npro_k > 1 (opr_partial.f90:185-195, 248-253)."""
        c = self.comm
        ut = trp_k_forward(c, {r: self.st[r]["txc"][src_idx][: self.n] for r in c.local_ranks}, self.npage, self.kmax)
        res = {}
        for r in c.local_ranks:
            res[r] = self.st[r]["rt"]
            self._partial(3, self.g[2], self.nlines, 1, self.nzt, ut[r], res[r])
        back = trp_k_backward(c, res, self.npage, self.kmax)
        for r in c.local_ranks:
            self.st[r]["txc"][dst_idx][: self.n].copy_(back[r])

    def poisson(self):
        """OPR_Poisson_FourierXZ_Factorize with the K-transposes of OPR_Fourier_Z_Forward/Backward (opr_fourier.f90:343-376,
        :394-428): forcing in tmp1 (txc[0]), Neumann data in hb/ht; returns p in tmp1 and dp/dy in tmp3 (txc[2])."""
        import torch
        c, L = self.comm, load()
        npage_c = self.nxh * self.ny
        nc = 2 * npage_c * self.kmax                      # doubles in a local spectral array
        for r in c.local_ranks:
            S = self.st[r]
            check(L.tlab_poisson_set_wall_planes(S["poisson"], _ptr(S["txc"][0]), _ptr(S["hb"]), _ptr(S["ht"])), "set_wall_planes")
            check(L.tlab_poisson_fft_x(S["poisson"], 1, _ptr(S["txc"][0]), _ptr(S["txc"][1])), "fft_x")          # p -> tmp2
        ct = trp_k_forward(c, {r: self.st[r]["txc"][1][:nc] for r in c.local_ranks}, npage_c, self.kmax, width=2)
        zf = {}
        for r in c.local_ranks:
            zf[r] = torch.empty_like(ct[r])
            check(L.tlab_poisson_fft_z(self.st[r]["poisson"], 1, _ptr(ct[r]), _ptr(zf[r])), "fft_z")
        fh = trp_k_backward(c, zf, npage_c, self.kmax, width=2)
        for r in c.local_ranks:
            S = self.st[r]
            S["txc"][3][:nc].copy_(fh[r])                                                                           # f^ -> tmp4
            check(L.tlab_poisson_ode(S["poisson"], _ptr(S["txc"][3]), _ptr(S["txc"][3]), _ptr(S["txc"][1])), "ode")  # p^ -> tmp4, dp^ -> tmp2
        for src, dst in ((3, 0), (1, 2)):                                                                            # p -> tmp1, dpdy -> tmp3
            ct = trp_k_forward(c, {r: self.st[r]["txc"][src][:nc] for r in c.local_ranks}, npage_c, self.kmax, width=2)
            zb = {}
            for r in c.local_ranks:
                zb[r] = torch.empty_like(ct[r])
                check(L.tlab_poisson_fft_z(self.st[r]["poisson"], -1, _ptr(ct[r]), _ptr(zb[r])), "fft_z")
            bk = trp_k_backward(c, zb, npage_c, self.kmax, width=2)
            for r in c.local_ranks:
                S = self.st[r]
                S["txc"][src][:nc].copy_(bk[r])
                check(L.tlab_poisson_fft_x(S["poisson"], -1, _ptr(S["txc"][src]), _ptr(S["txc"][dst])), "fft_x")

    # ---- the RHS ---------------------------------------------------------------------------------------------------------
    def RHS_GLOBAL_INCOMPRESSIBLE_1(self, dte):
        _use_torch_stream()
        L = load()
        nx, ny, kmax, n = self.nx, self.ny, self.kmax, self.n
        gx, gy = self.g[0], self.g[1]
        nu = self.visc
        T = lambda S, i: S["txc"][i]          # noqa: E731

        def add3(h, a, b, c_):
            check(L.tlab_pw_add3(_ptr(h), _ptr(a), _ptr(b), _ptr(c_), n), "add3")

        self._local(lambda r, S: self._burgers(1, gx, 0, nx, ny, kmax, nu, S["q"][0], S["q"][0], T(S, 0), T(S, 3)))       # :98
        self._local(lambda r, S: self._burgers(2, gy, 0, nx, ny, kmax, nu, S["q"][1], S["q"][1], T(S, 1), T(S, 4)))       # :99
        self.burgers_z(nu, lambda S: S["q"][2], 2, self_vel=True)                                                          # :100
        self._local(lambda r, S: self._burgers(2, gy, 1, nx, ny, kmax, nu, S["q"][0], S["q"][1], T(S, 6), T(S, 8)))       # :103
        self.burgers_z(nu, lambda S: S["q"][0], 7)                                                                         # :104
        self._local(lambda r, S: add3(S["hq"][0], T(S, 0), T(S, 6), T(S, 7)))
        self._local(lambda r, S: self._burgers(1, gx, 1, nx, ny, kmax, nu, S["q"][1], S["q"][0], T(S, 6), T(S, 8)))       # :115
        self.burgers_z(nu, lambda S: S["q"][1], 7)                                                                         # :116
        self._local(lambda r, S: add3(S["hq"][1], T(S, 1), T(S, 6), T(S, 7)))
        self._local(lambda r, S: self._burgers(1, gx, 1, nx, ny, kmax, nu, S["q"][2], S["q"][0], T(S, 6), T(S, 8)))       # :127
        self._local(lambda r, S: self._burgers(2, gy, 1, nx, ny, kmax, nu, S["q"][2], S["q"][1], T(S, 7), T(S, 8)))       # :128
        self._local(lambda r, S: add3(S["hq"][2], T(S, 2), T(S, 6), T(S, 7)))
        for i in range(self.nscal):                                                                                        # :149-162
            kap = self.visc / self.schmidt[i]
            self._local(lambda r, S: self._burgers(1, gx, 1, nx, ny, kmax, kap, S["s"][i], S["q"][0], T(S, 0), T(S, 8)))
            self._local(lambda r, S: self._burgers(2, gy, 1, nx, ny, kmax, kap, S["s"][i], S["q"][1], T(S, 1), T(S, 8)))
            self.burgers_z(kap, lambda S: S["s"][i], 2)
            self._local(lambda r, S: add3(S["hs"][i], T(S, 0), T(S, 1), T(S, 2)))
        # pressure (:188-260)
        self._local(lambda r, S: check(L.tlab_pw_axpy3(_ptr(T(S, 1)), _ptr(T(S, 2)), _ptr(T(S, 3)), _ptr(S["hq"][1]), _ptr(S["hq"][0]),
                                                       _ptr(S["hq"][2]), _ptr(S["q"][1]), _ptr(S["q"][0]), _ptr(S["q"][2]), 1.0 / dte, n), "axpy3"))
        self._local(lambda r, S: self._partial(2, gy, nx, ny, kmax, T(S, 1), T(S, 0)))                                     # :228
        self._local(lambda r, S: self._partial(1, gx, nx, ny, kmax, T(S, 2), T(S, 1)))                                     # :229
        self.partial_z(3, 2)                                                                                               # :230
        self._local(lambda r, S: check(L.tlab_pw_sum3(_ptr(T(S, 0)), _ptr(T(S, 1)), _ptr(T(S, 2)), n), "sum3"))
        self._local(lambda r, S: check(L.tlab_pw_get_wall_planes(_ptr(S["hq"][1]), _ptr(S["hb"]), _ptr(S["ht"]), nx, ny, kmax), "walls"))
        self.poisson()                                                                                                     # :284
        self._local(lambda r, S: self._partial(1, gx, nx, ny, kmax, T(S, 0), T(S, 1)))                                     # :319
        self.partial_z(0, 3)                                                                                               # :320
        self._local(lambda r, S: check(L.tlab_pw_sub3(_ptr(S["hq"][0]), _ptr(S["hq"][1]), _ptr(S["hq"][2]), _ptr(T(S, 1)), _ptr(T(S, 2)),
                                                      _ptr(T(S, 3)), n), "sub3"))
        # boundary conditions (:360-398); y is local to a z-slab, so BOUNDARY_BCS_NEUMANN_Y needs no communication
        types = list(zip(self.flow_jmin, self.flow_jmax)) + list(zip(self.scal_jmin, self.scal_jmax))

        def walls(r, S):
            for h, (tmin, tmax) in zip(S["hq"] + S["hs"], types):
                ibc = (1 if tmin == DNS_BCS_NEUMANN else 0) + (2 if tmax == DNS_BCS_NEUMANN else 0)
                if ibc:
                    check(L.tlab_boundary_bcs_neumann_y(gy._h, ibc, nx, ny, kmax, _ptr(h), _ptr(S["hb"]), _ptr(S["ht"]), _ptr(T(S, 0))), "bcs_neumann_y")
                check(L.tlab_pw_set_wall_planes(_ptr(h), _ptr(S["hb"]) if ibc & 1 else None, _ptr(S["ht"]) if ibc & 2 else None, nx, ny, kmax), "walls")
        self._local(walls)

    def set_bcs(self, velocity_jmin="noslip", velocity_jmax="noslip", scalar_jmin="dirichlet", scalar_jmax="dirichlet"):
        """As Dns.set_bcs (dns.ini [BoundaryConditions] keywords, boundary_bcs.f90:102-190)."""
        fj0, fj1, sj0, sj1 = _bcs_arrays(self.nscal, velocity_jmin, velocity_jmax, scalar_jmin, scalar_jmax)
        if fj0[1] != DNS_BCS_DIRICHLET or fj1[1] != DNS_BCS_DIRICHLET:
            raise TlabError("the wall-normal velocity must be Dirichlet")
        self.flow_jmin, self.flow_jmax = list(fj0), list(fj1)
        self.scal_jmin, self.scal_jmax = list(sj0)[: self.nscal], list(sj1)[: self.nscal]

    def TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(self, dte, kco=1.0, scale_tendencies=False):
        self.RHS_GLOBAL_INCOMPRESSIBLE_1(dte)
        L = load()
        for r in self.comm.local_ranks:
            S = self.st[r]
            for qf, hf in zip(S["q"] + S["s"], S["hq"] + S["hs"]):
                check(L.tlab_pw_rk_update(_ptr(qf), _ptr(hf), float(dte), float(kco), int(scale_tendencies), self.n), "rk_update")

    def substep_of_cycle(self, k, dtime):
        """k-th substep of consecutive RK steps (zeroes the tendencies at the start of each step, time.f90:212-216)."""
        s = k % self.rkm_endstep
        if s == 0:
            for r in self.comm.local_ranks:
                for t in self.st[r]["hq"] + self.st[r]["hs"]:
                    t.zero_()
        last = s == self.rkm_endstep - 1
        self.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * self.kdt[s], 1.0 if last else self.kco[s], not last)

    # ---- scatter / gather of global fields (tests, initial conditions) ---------------------------------------------------------
    def scatter(self, name, idx, global_field):
        """global_field: flat tensor nx*ny*nz_total (x fastest); every local rank takes its planes."""
        for r in self.comm.local_ranks:
            self.st[r][name][idx].copy_(global_field[r * self.n:(r + 1) * self.n])

    def gather_local(self, name, idx):
        """{rank: local slab} of the ranks simulated / owned here."""
        return {r: self.st[r][name][idx] for r in self.comm.local_ranks}

    def __del__(self):
        try:
            for S in self.st.values():
                load().tlab_poisson_plan_destroy(S["poisson"])
        except Exception:
            pass
