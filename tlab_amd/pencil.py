"""x/z pencil decomposition of the hot path: ims_npro_i x ims_npro_k ranks (BASELINE configs[3] "x/z-decomposed").

Mirrors the reference's cartesian layout (TLabMPI_Initialize, base/tlab_mpi_procs.f90:76-94):

    ims_pro_i = mod(ims_pro, ims_npro_i),  ims_pro_k = ims_pro / ims_npro_i
    ims_comm_x = the ranks of equal ims_pro_k,  ims_comm_z = the ranks of equal ims_pro_i

Rank (pro_i, pro_k) owns the block (imax, jmax, kmax) = (nx/npro_i, ny, nz/npro_k) of every field, x fastest.

  * x-operators (OPR_Partial_X, OPR_Burgers_X with ims_npro_i > 1: operators/opr_partial.f90:117-136, physics/opr_burgers.f90:236-262):
    TLabMPI_Trp_ExecI_Forward inside ims_comm_x -> complete x lines b(nx, nlines), nlines = jmax*kmax/npro_i -> the dir = 1 kernels on the
    box (nx, nlines, 1) -> TLabMPI_Trp_ExecI_Backward.  The transposed advecting velocity is kept and reused like the reference's tmp4.
  * z-operators: TLabMPI_Trp_ExecK_* inside ims_comm_z exactly as in the 1 x N driver (parallel.py, zmode = "transpose").
  * y-operators are local.
  * OPR_Poisson: the reference transposes the "extended" complex lines (imax/2 + 1 per rank, operators/opr_fourier.f90:131-134, 232-262)
    back to x-blocks and then K-transposes them for the z transform.  Here the I-transposition of the REAL forcing already leaves every rank
    with complete x lines of jmax*kmax/npro_i consecutive (j, k) lines -- when npro_i divides kmax that is a z-slab of kmax/npro_i planes,
    and ims_pro = pro_i + npro_i*pro_k orders those slabs along z.  From there the solver is the kx-pencil one of the 1 x N driver over ALL
    ranks (x-FFT on the slab, ONE all-to-all to kx-pencils, z-FFT + per-mode ODEs + inverse z-FFT on the pencil, one all-to-all back per
    output field, inverse x-FFT), and the two results return to x-blocks with one I-transposition each: 3 real I-transpositions and
    3 complex all-to-alls instead of the reference's 6 complex I- and 6 complex K-transpositions.

All arithmetic is the HIP library's (C ABI); the exchanges are `all_to_all`s of a communicator object (torch.distributed over RCCL / gloo,
or the in-process loopback used to verify the complete decomposed algorithm against the single domain on one GPU).
"""
import ctypes
import numpy as np

from .lib import load, check, TlabError, c_vp
from .operators import FdmPlan, _use_torch_stream, _ptr
from .dns import rk_coefficients, RKM_EXP3, DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN, _bcs_arrays
from .parallel import LoopbackComm, DistComm, trp_k_forward, trp_k_backward, trp_i_forward, trp_i_backward


class GroupComm:
    """A direction communicator (ims_comm_x or ims_comm_z) with every rank simulated in this process: groups = lists of world ranks in the
    order of their rank inside the communicator.  Tensors are keyed by WORLD rank."""

    def __init__(self, groups):
        self.groups = [list(g) for g in groups]
        self.size = len(self.groups[0])
        self.local_ranks = sorted(r for g in self.groups for r in g)
        self.rank = 0

    def all_to_all(self, sends):
        import torch
        out = {}
        for g in self.groups:
            for a, ra in enumerate(g):
                out[ra] = torch.stack([sends[rb][a] for rb in g])
        return out


def cart_groups(npro_i, npro_k):
    """(x groups, z groups) of world ranks (tlab_mpi_procs.f90:76-94)."""
    gx = [[pk * npro_i + pi for pi in range(npro_i)] for pk in range(npro_k)]
    gz = [[pk * npro_i + pi for pk in range(npro_k)] for pi in range(npro_i)]
    return gx, gz


def loopback_comms(npro_i, npro_k):
    gx, gz = cart_groups(npro_i, npro_k)
    return LoopbackComm(npro_i * npro_k), GroupComm(gx), GroupComm(gz)


def dist_comms(npro_i, npro_k):
    """World, x and z communicators over torch.distributed (every rank creates every group, as new_group requires)."""
    import torch.distributed as dist
    me = dist.get_rank()
    if dist.get_world_size() != npro_i * npro_k:
        raise TlabError("world size must be npro_i * npro_k")
    gx, gz = cart_groups(npro_i, npro_k)
    cx = cz = None
    for g in gx:
        h = dist.new_group(g)
        if me in g:
            cx = DistComm(h, key=me)
    for g in gz:
        h = dist.new_group(g)
        if me in g:
            cz = DistComm(h, key=me)
    return DistComm(None, key=me), cx, cz


class PencilDns:
    """RHS_GLOBAL_INCOMPRESSIBLE_1 + TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT on npro_i x npro_k blocks: the reference's operator sequence
    (tools/dns/rhs_global_incompressible_1.f90:98-375) with its transposed-velocity reuse (tmp4 / tmp6: :98-104, :115, :127, :152)."""

    def __init__(self, comms, npro_i, npro_k, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3,
                 hyper_bc1_ext=0.0, device="cuda"):
        import torch
        self.world, self.cx, self.cz = comms
        self.npi, self.npk = int(npro_i), int(npro_k)
        P = self.npi * self.npk
        if self.world.size != P or self.cx.size != self.npi or self.cz.size != self.npk:
            raise TlabError("communicator sizes do not match npro_i x npro_k")
        self.nx, self.ny, self.nzt = len(x), len(y), len(z)
        if self.nx % self.npi or self.nzt % self.npk:
            raise TlabError("nx, nz must be divisible by npro_i, npro_k")
        self.imax, self.kmax = self.nx // self.npi, self.nzt // self.npk
        if self.imax % 2:
            raise TlabError("imax must be even (opr_fourier.f90:73-76)")
        if self.kmax % self.npi:
            raise TlabError("npro_i must divide kmax: the x lines of a rank after the I-transposition form whole z planes")
        if (self.imax * self.ny) % self.npk:
            raise TlabError("imax*jmax must be divisible by npro_k (tlab_mpi_transpose.f90:292)")
        self.kmax2 = self.kmax // self.npi                 # planes per rank of the 1 x (npro_i npro_k) slabs the Poisson solver works on
        self.n = self.imax * self.ny * self.kmax
        self.npage_i, self.nlx = self.ny * self.kmax, self.ny * self.kmax2
        self.npage_k, self.nlz = self.imax * self.ny, self.imax * self.ny // self.npk
        self.nxh = self.nx // 2 + 1
        self.nscal, self.visc = int(nscal), float(visc)
        self.schmidt = [float(v) for v in schmidt][: self.nscal]
        self.g = [FdmPlan(x, True, True, hyper_bc1_ext=hyper_bc1_ext), FdmPlan(y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext),
                  FdmPlan(z, True, True, hyper_bc1_ext=hyper_bc1_ext)]
        self.kdt, self.kco = rk_coefficients(rkm_mode)
        self.rkm_endstep = len(self.kdt)
        self.flow_jmin, self.flow_jmax = [DNS_BCS_DIRICHLET] * 3, [DNS_BCS_DIRICHLET] * 3
        self.scal_jmin, self.scal_jmax = [DNS_BCS_DIRICHLET] * self.nscal, [DNS_BCS_DIRICHLET] * self.nscal
        self.isize_txc = (self.nx + 2) * self.ny * self.kmax2         # >= n; holds the complex slab (nx/2+1, ny, kmax2)
        base, rem = divmod(self.nxh, P)                               # kx ranges of the pencils
        self.nxl = [base + (1 if r < rem else 0) for r in range(P)]
        self.ioff = [r * base + min(r, rem) for r in range(P)]
        if min(self.nxl) < 1:
            raise TlabError("fewer kx modes than ranks")
        self._ioff_c = (ctypes.c_int * P)(*self.ioff)
        L = load()
        self.st = {}
        for r in self.world.local_ranks:
            z0 = lambda m: torch.zeros(m, dtype=torch.float64, device=device)   # noqa: E731
            h = c_vp(0)
            check(L.tlab_poisson_plan_create_pencil(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny, self.kmax2,
                                                    self.nzt, self.ioff[r], self.nxl[r]), "tlab_poisson_plan_create_pencil")
            S = dict(poisson=h, hb=z0(self.imax * self.kmax), ht=z0(self.imax * self.kmax), rt=z0(self.n),
                     pen=[z0(2 * self.nxl[r] * self.ny * self.nzt) for _ in range(3)], pack=[z0(2 * self.nxh * self.ny * self.kmax2) for _ in range(2)])
            for name, cnt, m in (("q", 3, self.n), ("s", self.nscal, self.n), ("hq", 3, self.n), ("hs", self.nscal, self.n), ("txc", 9, self.isize_txc)):
                S[name] = [z0(m) for _ in range(cnt)]
            self.st[r] = S

    def pro(self, r):
        """(ims_pro_i, ims_pro_k) of world rank r"""
        return r % self.npi, r // self.npi

    def _local(self, fn):
        for r in self.world.local_ranks:
            fn(r, self.st[r])

    def _burgers(self, d, ivel, nx, ny, nz, nu, s, u, res, tmp):
        check(load().tlab_opr_burgers(d, self.g[d - 1]._h, ivel, nx, ny, nz, 0, float(nu), _ptr(s), _ptr(u), _ptr(res), _ptr(tmp), 0), "tlab_opr_burgers")

    def _partial(self, d, nx, ny, nz, u, res):
        check(load().tlab_opr_partial(d, self.g[d - 1]._h, 1, nx, ny, nz, 0, _ptr(u), _ptr(res), c_vp(0)), "tlab_opr_partial")

    # ---- operators across the decomposed directions ------------------------------------------------------------------------------------
    def _burgers_t(self, d, nu, s_of, res_idx, self_vel):
        """OPR_Burgers_X (d = 1) / OPR_Burgers_Z (d = 3) through the I- / K-transposition; the advecting velocity is u / w."""
        if d == 1:
            comm, fwd, bwd, args, key, box = self.cx, trp_i_forward, trp_i_backward, (self.imax, self.npage_i), "u_t", (self.nx, self.nlx, 1)
        else:
            comm, fwd, bwd, args, key, box = self.cz, trp_k_forward, trp_k_backward, (self.npage_k, self.kmax), "w_t", (self.nlz, 1, self.nzt)
        if comm.size == 1:      # not decomposed in this direction
            bx = (self.imax, self.ny, self.kmax)
            self._local(lambda r, S: self._burgers(d, 0 if self_vel else 1, *bx, nu, s_of(S), S["q"][d - 1], S["txc"][res_idx], S["txc"][8]))
            return
        st = fwd(comm, {r: s_of(self.st[r])[: self.n] for r in comm.local_ranks}, *args)
        res = {}
        for r in comm.local_ranks:
            S = self.st[r]
            if self_vel:
                S[key] = st[r]
            res[r] = S["rt"]
            self._burgers(d, 0 if self_vel else 1, *box, nu, st[r], S[key], res[r], S["txc"][8][: self.n])
        back = bwd(comm, res, *args)
        for r in comm.local_ranks:
            self.st[r]["txc"][res_idx][: self.n].copy_(back[r])

    def _partial_t(self, d, src_idx, dst_idx):
        """OPR_Partial_X / _Z (OPR_P1) through the transposition (opr_partial.f90:117-136, 185-195)."""
        if d == 1:
            comm, fwd, bwd, args, box = self.cx, trp_i_forward, trp_i_backward, (self.imax, self.npage_i), (self.nx, self.nlx, 1)
        else:
            comm, fwd, bwd, args, box = self.cz, trp_k_forward, trp_k_backward, (self.npage_k, self.kmax), (self.nlz, 1, self.nzt)
        if comm.size == 1:
            self._local(lambda r, S: self._partial(d, self.imax, self.ny, self.kmax, S["txc"][src_idx], S["txc"][dst_idx]))
            return
        ut = fwd(comm, {r: self.st[r]["txc"][src_idx][: self.n] for r in comm.local_ranks}, *args)
        res = {}
        for r in comm.local_ranks:
            res[r] = self.st[r]["rt"]
            self._partial(d, *box, ut[r], res[r])
        back = bwd(comm, res, *args)
        for r in comm.local_ranks:
            self.st[r]["txc"][dst_idx][: self.n].copy_(back[r])

    # ---- Poisson ---------------------------------------------------------------------------------------------------------------------------
    def _to_slab(self, idx):
        """block txc[idx] (imax, ny, kmax) -> z-slab (nx, ny, kmax2) of the rank, in place of txc[idx]"""
        if self.npi == 1:
            return
        b = trp_i_forward(self.cx, {r: self.st[r]["txc"][idx][: self.n] for r in self.cx.local_ranks}, self.imax, self.npage_i)
        for r in self.cx.local_ranks:
            self.st[r]["txc"][idx][: self.n].copy_(b[r])

    def _to_block(self, idx):
        if self.npi == 1:
            return
        a = trp_i_backward(self.cx, {r: self.st[r]["txc"][idx][: self.n] for r in self.cx.local_ranks}, self.imax, self.npage_i)
        for r in self.cx.local_ranks:
            self.st[r]["txc"][idx][: self.n].copy_(a[r])

    def _pencil_exchange(self, forward, pen_idx, pack_idx):
        c, P = self.world, self.world.size
        send, scnt, recv, rcnt = {}, {}, {}, {}
        for r in c.local_ranks:
            S = self.st[r]
            mine = [2 * self.nxl[r] * self.ny * self.kmax2] * P
            peers = [2 * self.nxl[p] * self.ny * self.kmax2 for p in range(P)]
            if forward:
                send[r], scnt[r], recv[r], rcnt[r] = S["pack"][pack_idx], peers, S["pen"][pen_idx], mine
            else:
                send[r], scnt[r], recv[r], rcnt[r] = S["pen"][pen_idx], mine, S["pack"][pack_idx], peers
        return c.all_to_all_v(send, scnt, recv, rcnt)

    def _repack(self, S, slab, pack_idx, direction):
        P = self.world.size
        buf = S["pack"][pack_idx]
        if P <= 8:
            check(load().tlab_pencil_repack(_ptr(slab), _ptr(buf), self.nxh, self.ny, self.kmax2, P, self._ioff_c, direction), "tlab_pencil_repack")
            return
        a = slab[:2 * self.nxh * self.ny * self.kmax2].view(self.kmax2, self.ny, self.nxh, 2)
        off = 0
        for p in range(P):
            m = 2 * self.nxl[p] * self.ny * self.kmax2
            blk = buf[off:off + m].view(self.kmax2, self.ny, self.nxl[p], 2)
            if direction > 0:
                blk.copy_(a[:, :, self.ioff[p]:self.ioff[p] + self.nxl[p], :])
            else:
                a[:, :, self.ioff[p]:self.ioff[p] + self.nxl[p], :].copy_(blk)
            off += m

    def poisson(self):
        """OPR_Poisson(.., BCS_NN, ..): forcing in tmp1 (txc[0]), Neumann data in hb / ht; p -> tmp1, dp/dy -> tmp3 (txc[2])."""
        L = load()
        self._local(lambda r, S: check(L.tlab_pw_set_wall_planes(_ptr(S["txc"][0]), _ptr(S["hb"]), _ptr(S["ht"]), self.imax, self.ny, self.kmax),
                                       "wall planes"))                     # the Neumann data travel in the wall rows of the forcing (opr_elliptic.f90:310-311)
        self._to_slab(0)
        for r in self.world.local_ranks:
            S = self.st[r]
            check(L.tlab_poisson_fft_x(S["poisson"], 1, _ptr(S["txc"][0]), _ptr(S["txc"][1])), "fft_x")
            self._repack(S, S["txc"][1], 0, 1)
        self._pencil_exchange(True, 0, 0).wait()
        for r in self.world.local_ranks:
            S = self.st[r]
            b0, b1, b2 = S["pen"]
            check(L.tlab_poisson_fft_z(S["poisson"], 1, _ptr(b0), _ptr(b1)), "fft_z")
            check(L.tlab_poisson_ode(S["poisson"], _ptr(b1), _ptr(b1), _ptr(b2)), "ode")
            check(L.tlab_poisson_fft_z(S["poisson"], -1, _ptr(b1), _ptr(b0)), "fft_z")
            check(L.tlab_poisson_fft_z(S["poisson"], -1, _ptr(b2), _ptr(b1)), "fft_z")
        for pen_idx, dst in ((0, 0), (1, 2)):
            self._pencil_exchange(False, pen_idx, 0).wait()
            for r in self.world.local_ranks:
                S = self.st[r]
                self._repack(S, S["txc"][1], 0, -1)
                check(L.tlab_poisson_fft_x(S["poisson"], -1, _ptr(S["txc"][1]), _ptr(S["txc"][dst])), "fft_x")
            self._to_block(dst)

    # ---- the substep -------------------------------------------------------------------------------------------------------------------
    def RHS_GLOBAL_INCOMPRESSIBLE_1(self, dte):
        _use_torch_stream()
        L = load()
        nx, ny, kmax, n = self.imax, self.ny, self.kmax, self.n
        nu = self.visc
        T = lambda S, i: S["txc"][i]          # noqa: E731
        U = lambda i: (lambda S: S["q"][i])   # noqa: E731

        def add3(h, a, b, c_):
            check(L.tlab_pw_add3(_ptr(h), _ptr(a), _ptr(b), _ptr(c_), n), "add3")

        def burgers_y(ivel, nu_, s_of, res_idx):
            self._local(lambda r, S: self._burgers(2, ivel, nx, ny, kmax, nu_, s_of(S), S["q"][1], T(S, res_idx), T(S, 8)))

        self._burgers_t(1, nu, U(0), 0, True)                                           # :98   tmp1, u transposed kept
        burgers_y(0, nu, U(1), 1)                                                       # :99
        self._burgers_t(3, nu, U(2), 2, True)                                           # :100  tmp3, w transposed kept
        burgers_y(1, nu, U(0), 6)                                                       # :103
        self._burgers_t(3, nu, U(0), 7, False)                                          # :104
        self._local(lambda r, S: add3(S["hq"][0], T(S, 0), T(S, 6), T(S, 7)))
        self._burgers_t(1, nu, U(1), 6, False)                                          # :115
        self._burgers_t(3, nu, U(1), 7, False)                                          # :116
        self._local(lambda r, S: add3(S["hq"][1], T(S, 1), T(S, 6), T(S, 7)))
        self._burgers_t(1, nu, U(2), 6, False)                                          # :127
        burgers_y(1, nu, U(2), 7)                                                       # :128
        self._local(lambda r, S: add3(S["hq"][2], T(S, 2), T(S, 6), T(S, 7)))
        for i in range(self.nscal):                                                     # :149-162
            kap = self.visc / self.schmidt[i]
            sc = (lambda k: (lambda S: S["s"][k]))(i)
            self._burgers_t(1, kap, sc, 0, False)
            burgers_y(1, kap, sc, 1)
            self._burgers_t(3, kap, sc, 2, False)
            self._local(lambda r, S: add3(S["hs"][i], T(S, 0), T(S, 1), T(S, 2)))
        # pressure (:188-260)
        self._local(lambda r, S: check(L.tlab_pw_axpy3(_ptr(T(S, 1)), _ptr(T(S, 2)), _ptr(T(S, 3)), _ptr(S["hq"][1]), _ptr(S["hq"][0]),
                                                       _ptr(S["hq"][2]), _ptr(S["q"][1]), _ptr(S["q"][0]), _ptr(S["q"][2]), 1.0 / dte, n), "axpy3"))
        self._local(lambda r, S: self._partial(2, nx, ny, kmax, T(S, 1), T(S, 0)))     # :228
        self._partial_t(1, 2, 1)                                                        # :229
        self._partial_t(3, 3, 2)                                                        # :230
        self._local(lambda r, S: check(L.tlab_pw_sum3(_ptr(T(S, 0)), _ptr(T(S, 1)), _ptr(T(S, 2)), n), "sum3"))
        self._local(lambda r, S: check(L.tlab_pw_get_wall_planes(_ptr(S["hq"][1]), _ptr(S["hb"]), _ptr(S["ht"]), nx, ny, kmax), "walls"))
        self.poisson()                                                                  # :284
        self._partial_t(1, 0, 1)                                                        # :319
        self._partial_t(3, 0, 3)                                                        # :320
        self._local(lambda r, S: check(L.tlab_pw_sub3(_ptr(S["hq"][0]), _ptr(S["hq"][1]), _ptr(S["hq"][2]), _ptr(T(S, 1)), _ptr(T(S, 2)),
                                                      _ptr(T(S, 3)), n), "sub3"))
        # boundary conditions (:360-398); y is never split, BOUNDARY_BCS_NEUMANN_Y needs no communication
        types = list(zip(self.flow_jmin, self.flow_jmax)) + list(zip(self.scal_jmin, self.scal_jmax))
        gy = self.g[1]

        def walls(r, S):
            for h, (tmin, tmax) in zip(S["hq"] + S["hs"], types):
                ibc = (1 if tmin == DNS_BCS_NEUMANN else 0) + (2 if tmax == DNS_BCS_NEUMANN else 0)
                if ibc:
                    check(L.tlab_boundary_bcs_neumann_y(gy._h, ibc, nx, ny, kmax, _ptr(h), _ptr(S["hb"]), _ptr(S["ht"]), _ptr(T(S, 0))), "bcs_neumann_y")
                check(L.tlab_pw_set_wall_planes(_ptr(h), _ptr(S["hb"]) if ibc & 1 else None, _ptr(S["ht"]) if ibc & 2 else None, nx, ny, kmax), "walls")
        self._local(walls)

    def set_bcs(self, velocity_jmin="noslip", velocity_jmax="noslip", scalar_jmin="dirichlet", scalar_jmax="dirichlet"):
        fj0, fj1, sj0, sj1 = _bcs_arrays(self.nscal, velocity_jmin, velocity_jmax, scalar_jmin, scalar_jmax)
        if fj0[1] != DNS_BCS_DIRICHLET or fj1[1] != DNS_BCS_DIRICHLET:
            raise TlabError("the wall-normal velocity must be Dirichlet")
        self.flow_jmin, self.flow_jmax = list(fj0), list(fj1)
        self.scal_jmin, self.scal_jmax = list(sj0)[: self.nscal], list(sj1)[: self.nscal]

    def TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(self, dte, kco=1.0, scale_tendencies=False):
        self.RHS_GLOBAL_INCOMPRESSIBLE_1(dte)
        L = load()
        for r in self.world.local_ranks:
            S = self.st[r]
            for qf, hf in zip(S["q"] + S["s"], S["hq"] + S["hs"]):
                check(L.tlab_pw_rk_update(_ptr(qf), _ptr(hf), float(dte), float(kco), int(scale_tendencies), self.n), "rk_update")

    def substep_of_cycle(self, k, dtime):
        s = k % self.rkm_endstep
        if s == 0:
            for r in self.world.local_ranks:
                for t in self.st[r]["hq"] + self.st[r]["hs"]:
                    t.zero_()
        last = s == self.rkm_endstep - 1
        self.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * self.kdt[s], 1.0 if last else self.kco[s], not last)

    # ---- scatter / gather of global fields (tests, initial conditions) -----------------------------------------------------------------
    def block_of(self, r, global_field):
        """the (imax, ny, kmax) block of world rank r out of a flat global field nx*ny*nz (x fastest)"""
        pi, pk = self.pro(r)
        g = global_field.view(self.nzt, self.ny, self.nx)
        return g[pk * self.kmax:(pk + 1) * self.kmax, :, pi * self.imax:(pi + 1) * self.imax]

    def scatter(self, name, idx, global_field):
        for r in self.world.local_ranks:
            self.st[r][name][idx].view(self.kmax, self.ny, self.imax).copy_(self.block_of(r, global_field))

    def gather_local(self, name, idx):
        return {r: self.st[r][name][idx] for r in self.world.local_ranks}

    def __del__(self):
        try:
            for S in self.st.values():
                load().tlab_poisson_plan_destroy(S["poisson"])
        except Exception:
            pass



# ---------------------------------------------------------------------------------------------------------------------------------------------------
# The NATIVE driver (tlab_pencil_dns_*, tlab_amd/csrc/pencil.cpp): the C++ port of PencilDns behind the C ABI -- what a Fortran / MPI host calls.
# ---------------------------------------------------------------------------------------------------------------------------------------------------
_PA2A_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_vp, ctypes.c_int, ctypes.POINTER(c_vp), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(c_vp),
                            ctypes.POINTER(ctypes.c_longlong))
_PWAIT_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_vp, ctypes.c_int)
_PRED_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int)
_PDESTROY_FN = ctypes.CFUNCTYPE(None, c_vp)


class PencilTransport(ctypes.Structure):
    """struct tlab_pencil_transport"""
    _fields_ = [("ctx", c_vp), ("npro_i", ctypes.c_int), ("npro_k", ctypes.c_int), ("nlocal", ctypes.c_int), ("first", ctypes.c_int),
                ("alltoallv_start", _PA2A_FN), ("wait", _PWAIT_FN), ("allreduce", _PRED_FN), ("destroy", _PDESTROY_FN)]


class NativePencilDns:
    """tlab_pencil_dns_* with the interface of PencilDns (st[rank][name][i] tensors, scatter, gather_local, substep_of_cycle).
    transport: "loopback" (all npro_i x npro_k ranks in this process) or "rccl" (torch.distributed initialised, one rank per process: the
    ncclUniqueId of the library's own communicators travels through the group's object broadcast)."""

    def __init__(self, transport, npro_i, npro_k, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3,
                 hyper_bc1_ext=0.0, device="cuda", group=None):
        import torch
        L = load()
        self.npi, self.npk = int(npro_i), int(npro_k)
        self._tr = PencilTransport()
        self._keep = None
        if transport == "loopback":
            check(L.tlab_pencil_transport_loopback(ctypes.byref(self._tr), self.npi, self.npk), "tlab_pencil_transport_loopback")
        elif transport == "rccl":
            import torch.distributed as dist
            from . import comm as C
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            box = [C.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            self._keep = C.NativeComm(box[0], world, rank, self.npi, self.npk)
            C.check(C.load().tlab_comm_pencil_transport(self._keep._h, ctypes.byref(self._tr)), "tlab_comm_pencil_transport")
        else:
            raise TlabError("transport: loopback or rccl")
        self.transport = transport
        self.nx, self.ny, self.nzt = len(x), len(y), len(z)
        self.nscal, self.visc = int(nscal), float(visc)
        self.schmidt = [float(v) for v in schmidt][: self.nscal]
        self.g = [FdmPlan(x, True, True, hyper_bc1_ext=hyper_bc1_ext), FdmPlan(y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext),
                  FdmPlan(z, True, True, hyper_bc1_ext=hyper_bc1_ext)]
        self.kdt, self.kco = rk_coefficients(rkm_mode)
        self.rkm_endstep = len(self.kdt)
        sc = np.ascontiguousarray(self.schmidt if self.nscal else [1.0], dtype=np.float64)
        self._h = c_vp(0)
        rc = L.tlab_pencil_dns_create(ctypes.byref(self._h), ctypes.byref(self._tr), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny, self.nzt,
                                      self.nscal, self.visc, sc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        if rc != 0:          # a refused configuration leaves the transport context with the caller (include/tlab_amd.h)
            if self._tr.destroy:
                self._tr.destroy(self._tr.ctx)
            if self._keep is not None:
                self._keep.close()
                self._keep = None
        check(rc, "tlab_pencil_dns_create")
        self.imax, self.kmax, self.kmax2 = (int(L.tlab_pencil_dns_info(self._h, w)) for w in (0, 1, 2))
        self.isize_txc = int(L.tlab_pencil_dns_info(self._h, 3))
        self.n = self.imax * self.ny * self.kmax
        first, nlocal = int(L.tlab_pencil_dns_info(self._h, 5)), int(L.tlab_pencil_dns_info(self._h, 4))
        self.local_ranks = list(range(first, first + nlocal))
        self.st = {}
        for l, r in enumerate(self.local_ranks):
            S = {name: [torch.zeros(m, dtype=torch.float64, device=device) for _ in range(cnt)] for name, cnt, m in
                 (("q", 3, self.n), ("s", self.nscal, self.n), ("hq", 3, self.n), ("hs", self.nscal, self.n), ("txc", 9, self.isize_txc))}
            arr = lambda ts: (c_vp * max(len(ts), 1))(*[t.data_ptr() for t in ts])       # noqa: E731
            check(L.tlab_pencil_dns_bind(self._h, l, arr(S["q"]), arr(S["s"]), arr(S["hq"]), arr(S["hs"]), arr(S["txc"])), "tlab_pencil_dns_bind")
            self.st[r] = S

    def redraw_arrays(self, pool=34, seed=0):
        """As NativeSlabDns.redraw_arrays (tlab_amd/placement.py)."""
        from .placement import redraw_rank_arrays
        return redraw_rank_arrays(self, "tlab_pencil_dns_bind", pool=pool, seed=seed)

    def pro(self, r):
        return r % self.npi, r // self.npi

    def set_bcs(self, velocity_jmin="noslip", velocity_jmax="noslip", scalar_jmin="dirichlet", scalar_jmax="dirichlet"):
        fj0, fj1, sj0, sj1 = _bcs_arrays(self.nscal, velocity_jmin, velocity_jmax, scalar_jmin, scalar_jmax)
        ia = lambda v, m: (ctypes.c_int * max(m, 1))(*list(v)[:m])       # noqa: E731
        check(load().tlab_pencil_dns_set_bcs(self._h, ia(fj0, 3), ia(fj1, 3), ia(sj0, self.nscal), ia(sj1, self.nscal)), "tlab_pencil_dns_set_bcs")

    def RHS_GLOBAL_INCOMPRESSIBLE_1(self, dte):
        _use_torch_stream()
        check(load().tlab_pencil_dns_rhs(self._h, float(dte)), "tlab_pencil_dns_rhs")

    def TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(self, dte, kco=1.0, scale_tendencies=False):
        _use_torch_stream()
        check(load().tlab_pencil_dns_substep(self._h, float(dte), float(kco), int(scale_tendencies)), "tlab_pencil_dns_substep")

    def substep_of_cycle(self, k, dtime):
        s = k % self.rkm_endstep
        if s == 0:
            _use_torch_stream()
            check(load().tlab_pencil_dns_begin_step(self._h), "tlab_pencil_dns_begin_step")
        last = s == self.rkm_endstep - 1
        self.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * self.kdt[s], 1.0 if last else self.kco[s], not last)

    def block_of(self, r, global_field):
        pi, pk = self.pro(r)
        g = global_field.view(self.nzt, self.ny, self.nx)
        return g[pk * self.kmax:(pk + 1) * self.kmax, :, pi * self.imax:(pi + 1) * self.imax]

    def scatter(self, name, idx, global_field):
        for r in self.local_ranks:
            self.st[r][name][idx].view(self.kmax, self.ny, self.imax).copy_(self.block_of(r, global_field))

    def gather_local(self, name, idx):
        return {r: self.st[r][name][idx] for r in self.local_ranks}

    def close(self):
        if self._h:
            load().tlab_pencil_dns_destroy(self._h)
            self._h = c_vp(0)
        if self._keep is not None:
            self._keep.close()
            self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass
