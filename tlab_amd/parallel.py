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

Two algorithms are built on that decomposition:

  zmode = "transpose": the reference's scheme.  Every z-operator and the z-FFT go through a K-transposition (12 + 6 field
      all-to-alls per substep); kept for thin slabs and as the cross-check of the second one.
  zmode = "halo" (default when the slabs are thick enough, kmax >~ 50): no field is transposed for a derivative.  The compact
      z-systems are partitioned at the slab boundaries (tlab_amd/csrc/zslab.hip): an operator needs 3 halo planes of its
      operand and one value per line and system from each neighbour -- two small point-to-point messages over the xGMI link
      to each neighbour, overlapped with the x/y operators.  The Poisson solver goes from the z-slab to a kx-pencil with ONE
      all-to-all after the x-FFT (z-FFT, ODEs and inverse z-FFT are then local) and back with one per output field:
      3 field all-to-alls per substep instead of 18.

`LoopbackComm` runs all npro_k ranks inside ONE process on one device, exchanging blocks by direct copies: the complete
decomposed algorithm (index maps, per-rank wavenumber offsets, singular-mode ownership) can then be verified against the
single-domain result on a single GPU (tests/test_gpu_slab.py) although only the driver's 8-GPU node can run it for real.
"""
import ctypes
import os
import numpy as np

from .lib import load, check, TlabError, c_vp
from .operators import FdmPlan, _use_torch_stream, _ptr
from .dns import rk_coefficients, RKM_EXP3, DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN, _bcs_arrays


# ------------------------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------------------------
class DistComm:
    """ims_comm_z over torch.distributed: one rank per process."""

    def __init__(self, group=None, key=None):
        """key: what the tensors of this process are filed under in the {rank: tensor} arguments (default: the rank inside the group; the
        pencil driver files everything under the WORLD rank, whichever communicator carries it)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.key = self.rank if key is None else key
        self.local_ranks = [self.key]

        # functional testing of the multi-process path on a box with fewer GPUs than ranks: gloo cannot move device memory in
        # every collective, so the payload is staged through the host (never used for measurements)
        self.stage_host = dist.get_backend(group) == "gloo"

    def all_to_all(self, sends):
        """sends: {rank: tensor [size, chunk]} (contiguous).  Returns {rank: tensor [size(src), chunk]}."""
        import torch
        (r, s), = sends.items()
        if self.stage_host and s.is_cuda:
            hs = s.cpu()
            ho = torch.empty_like(hs)
            self.dist.all_to_all_single(ho, hs, group=self.group)
            return {r: ho.to(s.device)}
        out = torch.empty_like(s)
        self.dist.all_to_all_single(out, s, group=self.group)
        return {r: out}

    def all_to_all_v(self, send, send_counts, recv, recv_counts):
        """send/recv: {rank: flat tensor}; *_counts: {rank: [elements per peer]}.  In place into recv.  Returns a waitable."""
        r = self.key
        if self.stage_host and send[r].is_cuda:
            import torch
            hs = send[r].cpu()
            ho = torch.empty(recv[r].numel(), dtype=hs.dtype)
            self.dist.all_to_all_single(ho, hs, output_split_sizes=list(recv_counts[r]), input_split_sizes=list(send_counts[r]), group=self.group)
            recv[r].copy_(ho)
            return _Done()
        w = self.dist.all_to_all_single(recv[r], send[r], output_split_sizes=list(recv_counts[r]), input_split_sizes=list(send_counts[r]),
                                        group=self.group, async_op=True)
        return _Works([w])

    def neighbor_exchange(self, to_left, to_right, from_right, from_left):
        """Periodic ring: to_left[r][i] lands in from_right[r-1][i], to_right[r][i] in from_left[r+1][i].  Lists of contiguous
        tensors per rank.  Posting order (sends: left then right; receives: from right then from left) keeps the pairing right
        when both neighbours are the same rank (2 ranks)."""
        dist, r, P = self.dist, self.key, self.size
        left, right = (self.rank - 1) % P, (self.rank + 1) % P
        if self.stage_host and (to_left[r] + to_right[r]) and (to_left[r] + to_right[r])[0].is_cuda:
            import torch
            sl, sr = [t.cpu() for t in to_left[r]], [t.cpu() for t in to_right[r]]
            rr, rl = [torch.empty(t.shape, dtype=t.dtype) for t in from_right[r]], [torch.empty(t.shape, dtype=t.dtype) for t in from_left[r]]
            ops = [dist.P2POp(dist.isend, t, left, self.group) for t in sl] + [dist.P2POp(dist.isend, t, right, self.group) for t in sr]
            ops += [dist.P2POp(dist.irecv, t, right, self.group) for t in rr] + [dist.P2POp(dist.irecv, t, left, self.group) for t in rl]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for d, h in zip(from_right[r] + from_left[r], rr + rl):
                d.copy_(h)
            return _Done()
        ops = [dist.P2POp(dist.isend, t, left, self.group) for t in to_left[r]] + [dist.P2POp(dist.isend, t, right, self.group) for t in to_right[r]]
        ops += [dist.P2POp(dist.irecv, t, right, self.group) for t in from_right[r]] + [dist.P2POp(dist.irecv, t, left, self.group) for t in from_left[r]]
        return _Works(dist.batch_isend_irecv(ops))


    def all_reduce(self, values, op="max"):
        """values: {rank: list of floats}.  MPI_ALLREDUCE(MPI_MAX / MPI_MIN) of a few scalars (time.f90:522, minmax.f90)."""
        import torch
        (r, v), = values.items()
        dev = "cpu" if self.stage_host or not torch.cuda.is_available() else "cuda"
        t = torch.tensor(list(v), dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.MIN, group=self.group)
        return [float(x) for x in t.cpu()]


class _Done:
    def wait(self):
        pass


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()          # NCCL: makes the current stream wait for the collective; does not block the host


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

    def all_to_all_v(self, send, send_counts, recv, recv_counts):
        P = self.size
        for dst in range(P):
            ro = 0
            for src in range(P):
                so = sum(send_counts[src][:dst])
                cnt = send_counts[src][dst]
                assert cnt == recv_counts[dst][src]
                recv[dst][ro:ro + cnt].copy_(send[src][so:so + cnt])
                ro += cnt
        return _Done()

    def neighbor_exchange(self, to_left, to_right, from_right, from_left):
        P = self.size
        for r in range(P):
            for t, d in zip(to_left[r], from_right[(r - 1) % P]):
                d.copy_(t)
            for t, d in zip(to_right[r], from_left[(r + 1) % P]):
                d.copy_(t)
        return _Done()


    def all_reduce(self, values, op="max"):
        f = max if op == "max" else min
        n = len(next(iter(values.values())))
        return [f(values[r][i] for r in self.local_ranks) for i in range(n)]


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


def trp_i_forward(comm, a, imax, npage, width=1):
    """TLabMPI_Trp_ExecI_Forward (tlab_mpi_transpose.f90:232-256, plan :205-230): local a(imax, npage) (x fastest, npage = jmax*kmax lines)
    -> b(imax*npro_i, nlines), nlines = npage/npro_i: peer p gets my lines [p*nlines, (p+1)*nlines) and stores my x-segment at x-offset
    my_rank*imax of each of its lines.  The send side needs no packing (a block of lines is contiguous), the receive side one strided copy.
    Not used by the 1 x N slab driver; completes the transposition layer of SURVEY 8a a14."""
    import torch
    P = comm.size
    if npage % P != 0:
        raise TlabError("I-transposition: npage must be divisible by the number of x pencils (tlab_mpi_transpose.f90:223)")
    nl = npage // P
    recv = comm.all_to_all({r: a[r].view(P, nl * imax * width) for r in comm.local_ranks})      # [src][line][x of src]
    out = {}
    for r in comm.local_ranks:
        b = torch.empty(nl, P, imax * width, dtype=a[r].dtype, device=a[r].device)
        b.copy_(recv[r].view(P, nl, imax * width).transpose(0, 1))
        out[r] = b.reshape(-1)                                                                 # b(x_global, line), x fastest
    return out


def trp_i_backward(comm, b, imax, npage, width=1):
    """TLabMPI_Trp_ExecI_Backward (tlab_mpi_transpose.f90:260-286): exact inverse of trp_i_forward."""
    P = comm.size
    nl = npage // P
    sends = {r: b[r].view(nl, P, imax * width).transpose(0, 1).contiguous().view(P, nl * imax * width) for r in comm.local_ranks}
    recv = comm.all_to_all(sends)                                                              # [src = owner of the lines][line][my x]
    return {r: recv[r].reshape(-1) for r in comm.local_ranks}


# ------------------------------------------------------------------------------------------------------------------
# the decomposed RK substep
# ------------------------------------------------------------------------------------------------------------------
def pencil_stage_layout(ioff, nxl, ny, kmax):
    """Block map of the two-stage pencil exchange (SlabDns._poisson_pencil_staged, tlab_pencil_repack_blocks): every rank's kx range
    [ioff[p], ioff[p] + nxl[p]) is cut into halves A (the first ceil(nxl/2) columns) and B; the pack buffer holds all A blocks (by rank) ahead of
    all B blocks.  Returns (start, base, split, nxa, nxb): block 2p / 2p+1 = half A / B of rank p starts at kx = start[.] and at element
    base[.] of the buffer, [kmax][ny][width] each; split = elements of the A part."""
    P = len(nxl)
    nxa = [(w + 1) // 2 for w in nxl]
    nxb = [w - a for w, a in zip(nxl, nxa)]
    start, base = [], []
    offa, offb = 0, sum(nxa) * ny * kmax
    for p in range(P):
        start += [ioff[p], ioff[p] + nxa[p]]
        base += [offa, offb]
        offa += nxa[p] * ny * kmax
        offb += nxb[p] * ny * kmax
    return start, base, sum(nxa) * ny * kmax, nxa, nxb


class SlabDns:
    """RHS_GLOBAL_INCOMPRESSIBLE_1 + TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT with ims_npro_k = comm.size.
    Same operator sequence as tlab_amd/csrc/rhs.cpp (= tools/dns/rhs_global_incompressible_1.f90:98-375); the z-operators
    go through the K-transposes, the transposed w is kept and reused like the reference does with tmp6 (:100,104,116,152)."""

    HALO = 3          # planes each side: the 7-diagonal right-hand side of the second derivative reaches 3 rows

    def __init__(self, comm, x, y, z, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3,
                 hyper_bc1_ext=0.0, device="cuda", zmode="auto", zchunk=0, plans=None, gy_elliptic=None):
        """plans: optional (gx, gy, gz) built elsewhere (FdmPlan.from_tables with a host's CompactDirect6 tables in y); gy_elliptic: the y plan of
        EllipticOrder = CompactDirect6 -> OPR_Poisson_FourierXZ_Direct on the slabs (the scheme set of examples/Case81-93)."""
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
        self.nlines = self.npage // P
        self.nscal, self.visc = int(nscal), float(visc)
        self.schmidt = [float(v) for v in schmidt][: self.nscal]
        self.g = list(plans) if plans is not None else [
            FdmPlan(x, True, True, hyper_bc1_ext=hyper_bc1_ext), FdmPlan(y, False, yuniform, hyper_bc1_ext=hyper_bc1_ext),
            FdmPlan(z, True, True, hyper_bc1_ext=hyper_bc1_ext)]
        self.gy_elliptic = gy_elliptic
        self.kdt, self.kco = rk_coefficients(rkm_mode)
        self.rkm_endstep = len(self.kdt)
        self._fresh = False
        self.flow_jmin, self.flow_jmax = [DNS_BCS_DIRICHLET] * 3, [DNS_BCS_DIRICHLET] * 3
        self.scal_jmin, self.scal_jmax = [DNS_BCS_DIRICHLET] * self.nscal, [DNS_BCS_DIRICHLET] * self.nscal
        self.isize_txc = (self.nx + 2) * self.ny * self.kmax
        L = load()
        if zmode not in ("auto", "halo", "transpose"):
            raise TlabError("zmode: auto, halo or transpose")
        # ---- z-slab operator plans (halo mode); a slab too thin for them sends everything to the transpose path ----
        zplans = {}
        if zmode != "transpose" and P > 1:
            for r in comm.local_ranks:
                h = c_vp(0)
                rc = L.tlab_zslab_plan_create(ctypes.byref(h), self.g[2]._h, self.kmax, r * self.kmax, int(zchunk))
                if rc != 0:
                    for hh in zplans.values():
                        L.tlab_zslab_plan_destroy(hh)
                    zplans = {}
                    if zmode == "halo":
                        check(rc, "tlab_zslab_plan_create")
                    break
                zplans[r] = h
        self.zmode = "halo" if zplans else "transpose"
        if self.zmode == "transpose" and (self.npage % P or (self.nxh * self.ny) % P):
            raise TlabError("imax*jmax and (imax/2+1)*jmax must be divisible by the number of z slabs (tlab_mpi_transpose.f90:292)")
        # kx ranges of the pencils: [ioff[r], ioff[r] + nxl[r])
        base, rem = divmod(self.nxh, P)
        self.nxl = [base + (1 if r < rem else 0) for r in range(P)]
        self.ioff = [r * base + min(r, rem) for r in range(P)]
        self._ioff_c = (ctypes.c_int * P)(*self.ioff)
        if self.zmode == "halo" and min(self.nxl) < 1:
            raise TlabError("fewer kx modes than ranks")
        # two-stage pencil Poisson (see _poisson_pencil_staged); TLAB_PENCIL_STAGES=1 keeps the one-piece exchange
        self.stages = 2 if (self.zmode == "halo" and gy_elliptic is None and min(self.nxl) >= 2 and 2 * P <= 16
                            and os.environ.get("TLAB_PENCIL_STAGES", "2") != "1") else 1
        self.nxa = [(w + 1) // 2 for w in self.nxl]
        Hn = self.HALO * self.npage
        self.st = {}
        for r in comm.local_ranks:
            def field(m):
                """m doubles with HALO planes of room on both sides: the slab's first plane is at ext[HALO*npage]."""
                ext = torch.zeros(m + 2 * Hn, dtype=torch.float64, device=device)
                return ext, ext[Hn:Hn + m]
            z0 = lambda m: torch.zeros(m, dtype=torch.float64, device=device)   # noqa: E731
            h = c_vp(0)
            if gy_elliptic is not None:
                mode, a, b = (1, self.ioff[r], self.nxl[r]) if self.zmode == "halo" else (0, r * self.kmax, P)
                check(L.tlab_poisson_plan_create_direct_decomposed(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                                                   self.kmax, self.nzt, mode, a, b, gy_elliptic._h),
                      "tlab_poisson_plan_create_direct_decomposed")
            elif self.zmode == "halo" and self.stages == 2:
                # the kx range of every rank in two halves, each with its own plan: the solves of one half hide the transfers of the other
                hb_ = c_vp(0)
                check(L.tlab_poisson_plan_create_pencil(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                                        self.kmax, self.nzt, self.ioff[r], self.nxa[r]), "tlab_poisson_plan_create_pencil")
                check(L.tlab_poisson_plan_create_pencil(ctypes.byref(hb_), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                                        self.kmax, self.nzt, self.ioff[r] + self.nxa[r], self.nxl[r] - self.nxa[r]),
                      "tlab_poisson_plan_create_pencil")
            elif self.zmode == "halo":
                check(L.tlab_poisson_plan_create_pencil(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                                        self.kmax, self.nzt, self.ioff[r], self.nxl[r]), "tlab_poisson_plan_create_pencil")
            else:
                check(L.tlab_poisson_plan_create_slab(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, self.nx, self.ny,
                                                      self.kmax, self.nzt, r * self.kmax, P), "tlab_poisson_plan_create_slab")
            S = dict(ext={}, rt=z0(self.n), hb=z0(self.nx * self.kmax), ht=z0(self.nx * self.kmax), poisson=h)
            if self.zmode == "halo" and self.stages == 2:
                S["poisson_b"] = hb_
            for name, cnt, m in (("q", 3, self.n), ("s", self.nscal, self.n), ("hq", 3, self.n), ("hs", self.nscal, self.n), ("txc", 9, self.isize_txc)):
                pairs = [field(m) for _ in range(cnt)]
                S["ext"][name] = [e for e, _ in pairs]
                S[name] = [i for _, i in pairs]
            if self.zmode == "halo":
                S["zplan"] = zplans[r]
                nmsg = 2 * (3 + self.nscal)                               # (first, second derivative) x fields of one batch
                for k in ("head", "tail", "head_right", "tail_left"):
                    S[k] = z0(nmsg * self.npage)
                npen = 2 * self.nxl[r] * self.ny * self.nzt               # doubles of a complex pencil (nxl, ny, nz_total)
                S["pen"] = [z0(npen) for _ in range(3)]
                S["pack"] = [z0(2 * self.nxh * self.ny * self.kmax) for _ in range(2)]
            self.st[r] = S

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
            if self.gy_elliptic is not None:
                check(L.tlab_poisson_direct_ode(S["poisson"], 3, _ptr(S["txc"][3]), _ptr(S["txc"][3])), "direct_ode")  # p^ -> tmp4 (BCS_NN)
            else:
                check(L.tlab_poisson_ode(S["poisson"], _ptr(S["txc"][3]), _ptr(S["txc"][3]), _ptr(S["txc"][1])), "ode")  # p^ -> tmp4, dp^ -> tmp2
        for src, dst in (((3, 0),) if self.gy_elliptic is not None else ((3, 0), (1, 2))):                          # p -> tmp1, dpdy -> tmp3
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
        if self.gy_elliptic is not None:     # dp/dy = OPR_Partial_Y(p) with the plan of the derivatives (opr_elliptic.f90:447-449)
            self._local(lambda r, S: self._partial(2, self.g[1], self.nx, self.ny, self.kmax, S["txc"][0], S["txc"][2]))

    # ---- halo mode: neighbour exchanges -----------------------------------------------------------------------------------
    def _halo_start(self, fields, nplanes=None):
        """Starts the exchange of the halo planes of fields = [(name, idx), ...]: my last planes go to the right neighbour's
        planes -H..-1, my first planes to the left neighbour's planes kmax..kmax+H-1 (periodic in z).  Zero-copy: the planes are
        contiguous in memory on both sides."""
        H = self.HALO if nplanes is None else nplanes
        Hn, Hfull, n = H * self.npage, self.HALO * self.npage, self.n
        tl, tr, fr, fl = {}, {}, {}, {}
        for r in self.comm.local_ranks:
            S = self.st[r]
            tl[r] = [S[nm][i][:Hn] for nm, i in fields]
            tr[r] = [S[nm][i][n - Hn:n] for nm, i in fields]
            fr[r] = [S["ext"][nm][i][Hfull + n:Hfull + n + Hn] for nm, i in fields]          # planes kmax .. kmax+H-1
            fl[r] = [S["ext"][nm][i][Hfull - Hn:Hfull] for nm, i in fields]                  # planes -H .. -1
        return self.comm.neighbor_exchange(tl, tr, fr, fl)

    def _msg_start(self, nrows):
        """head -> left neighbour (its head_right), tail -> right neighbour (its tail_left); nrows lines-sets of nx*ny."""
        m = nrows * self.npage
        c = self.comm
        return c.neighbor_exchange({r: [self.st[r]["head"][:m]] for r in c.local_ranks}, {r: [self.st[r]["tail"][:m]] for r in c.local_ranks},
                                   {r: [self.st[r]["head_right"][:m]] for r in c.local_ranks}, {r: [self.st[r]["tail_left"][:m]] for r in c.local_ranks})

    def _zburgers(self, phase, S, row, nu, s, vel, res):
        """Phase 1 / 2 of OPR_Burgers_Z on the slab; row = first message row (2 rows per call: first, second derivative)."""
        o = row * self.npage * 8
        P = lambda t: c_vp(t.data_ptr() + o)                 # noqa: E731
        check(load().tlab_zslab_burgers_z(S["zplan"], phase, self.nx, self.ny, float(nu), _ptr(s), _ptr(vel) if vel is not None else None,
                                          P(S["head"]), P(S["tail"]), P(S["tail_left"]), P(S["head_right"]),
                                          _ptr(res) if res is not None else None, 1), "tlab_zslab_burgers_z")

    def _zpartial(self, phase, S, u, ub, scale, res, acc):
        check(load().tlab_zslab_partial_z(S["zplan"], phase, self.nx, self.ny, _ptr(u), _ptr(ub) if ub is not None else None, float(scale),
                                          _ptr(S["head"]), _ptr(S["tail"]), _ptr(S["tail_left"]), _ptr(S["head_right"]),
                                          _ptr(res) if res is not None else None, int(acc)), "tlab_zslab_partial_z")

    def _pencil_forward(self, src_idx):
        """complex slab txc[src] (nxh, ny, kmax) of every rank -> pen[0] (nxl, ny, nz_total): ONE all-to-all; the receive side
        needs no unpacking because z is the slowest index of the pencil."""
        c, P = self.comm, self.comm.size
        send, scnt, recv, rcnt = {}, {}, {}, {}
        for r in c.local_ranks:
            S = self.st[r]
            buf = S["pack"][0]
            if P <= 8:      # one HIP kernel
                check(load().tlab_pencil_repack(_ptr(S["txc"][src_idx]), _ptr(buf), self.nxh, self.ny, self.kmax, P, self._ioff_c, 1), "tlab_pencil_repack")
            else:           # torch strided copies
                a = S["txc"][src_idx][:2 * self.nxh * self.ny * self.kmax].view(self.kmax, self.ny, self.nxh, 2)
                off = 0
                for p in range(P):
                    m = 2 * self.nxl[p] * self.ny * self.kmax
                    buf[off:off + m].view(self.kmax, self.ny, self.nxl[p], 2).copy_(a[:, :, self.ioff[p]:self.ioff[p] + self.nxl[p], :])
                    off += m
            cnts = [2 * self.nxl[p] * self.ny * self.kmax for p in range(P)]
            send[r], scnt[r] = buf, cnts
            recv[r], rcnt[r] = S["pen"][0], [2 * self.nxl[r] * self.ny * self.kmax] * P
        return c.all_to_all_v(send, scnt, recv, rcnt)

    def _pencil_backward_start(self, pen_idx, pack_idx):
        c, P = self.comm, self.comm.size
        send, scnt, recv, rcnt = {}, {}, {}, {}
        for r in c.local_ranks:
            S = self.st[r]
            send[r], scnt[r] = S["pen"][pen_idx], [2 * self.nxl[r] * self.ny * self.kmax] * P
            recv[r], rcnt[r] = S["pack"][pack_idx], [2 * self.nxl[p] * self.ny * self.kmax for p in range(P)]
        return c.all_to_all_v(send, scnt, recv, rcnt)

    def _pencil_backward_finish(self, pack_idx, dst_idx):
        P = self.comm.size
        for r in self.comm.local_ranks:
            S = self.st[r]
            buf = S["pack"][pack_idx]
            if P <= 8:
                check(load().tlab_pencil_repack(_ptr(S["txc"][dst_idx]), _ptr(buf), self.nxh, self.ny, self.kmax, P, self._ioff_c, -1), "tlab_pencil_repack")
                continue
            a = S["txc"][dst_idx][:2 * self.nxh * self.ny * self.kmax].view(self.kmax, self.ny, self.nxh, 2)
            off = 0
            for p in range(P):
                m = 2 * self.nxl[p] * self.ny * self.kmax
                a[:, :, self.ioff[p]:self.ioff[p] + self.nxl[p], :].copy_(buf[off:off + m].view(self.kmax, self.ny, self.nxl[p], 2))
                off += m

    def _poisson_pencil_staged(self):
        """The same solve with every rank's kx range cut in two halves A, B (plans S["poisson"], S["poisson_b"]).  The pack buffer holds all A blocks
        ahead of all B blocks, so each half is one all-to-all of its own; the collectives queue up in the order fwd A, fwd B, back p A, back dp A,
        back p B, back dp B and the z-transforms and per-mode solves of A run under fwd B, those of B under the returns of A.  Exposed: about
        half of one forward and of the two backward exchanges instead of all three.  Same kernels on the same modes: the result is that of
        _poisson_pencil to the bit."""
        L = load()
        c, P = self.comm, self.comm.size
        ny, kmax, nzt = self.ny, self.kmax, self.nzt
        nxa, nxb = self.nxa, [w - a for w, a in zip(self.nxl, self.nxa)]
        if not hasattr(self, "_stage_maps"):
            start, base, split_c, _, _ = pencil_stage_layout(self.ioff, self.nxl, ny, kmax)      # in complex elements
            self._stage_maps = ((ctypes.c_int * (2 * P))(*start), (ctypes.c_longlong * (2 * P))(*base), 2 * split_c)
        start, base, split = self._stage_maps                      # split: doubles of the A part of a pack buffer
        halves = (("poisson", nxa, 0), ("poisson_b", nxb, 1))

        def pen(S, i, h):
            """buffer i of half h of this rank's pencil (nx_half, ny, nz_total), carved out of pen[i]"""
            ma = 2 * nxa[S["rank"]] * ny * nzt
            return S["pen"][i][:ma] if h == 0 else S["pen"][i][ma:]

        def pack(S, i, h):
            return S["pack"][i][:split] if h == 0 else S["pack"][i][split:]

        def exchange(i_pen, i_pack, h, nxh_, forward):
            send, scnt, recv, rcnt = {}, {}, {}, {}
            for r in c.local_ranks:
                S = self.st[r]
                slab_side, cnt_slab = pack(S, i_pack, h), [2 * nxh_[p] * ny * kmax for p in range(P)]
                pen_side, cnt_pen = pen(S, i_pen, h), [2 * nxh_[r] * ny * kmax] * P
                if forward:
                    send[r], scnt[r], recv[r], rcnt[r] = slab_side, cnt_slab, pen_side, cnt_pen
                else:
                    send[r], scnt[r], recv[r], rcnt[r] = pen_side, cnt_pen, slab_side, cnt_slab
            return c.all_to_all_v(send, scnt, recv, rcnt)

        for r in c.local_ranks:
            S = self.st[r]
            S["rank"] = r
            check(L.tlab_poisson_set_wall_planes(S["poisson"], _ptr(S["txc"][0]), _ptr(S["hb"]), _ptr(S["ht"])), "set_wall_planes")
            check(L.tlab_poisson_fft_x(S["poisson"], 1, _ptr(S["txc"][0]), _ptr(S["txc"][1])), "fft_x")
            check(L.tlab_pencil_repack_blocks(_ptr(S["txc"][1]), _ptr(S["pack"][0]), self.nxh, ny, kmax, 2 * P, start, base, 1), "tlab_pencil_repack_blocks")
        fwd = [exchange(0, 0, h, nx_, True) for _, nx_, h in halves]
        back = []
        for key, nx_, h in halves:
            fwd[h].wait()
            for r in c.local_ranks:
                S = self.st[r]
                b0, b1, b2 = pen(S, 0, h), pen(S, 1, h), pen(S, 2, h)
                check(L.tlab_poisson_fft_z(S[key], 1, _ptr(b0), _ptr(b1)), "fft_z")
                check(L.tlab_poisson_ode(S[key], _ptr(b1), _ptr(b1), _ptr(b2)), "ode")
                check(L.tlab_poisson_fft_z(S[key], -1, _ptr(b1), _ptr(b0)), "fft_z")
            wp = exchange(0, 0, h, nx_, False)                       # p^ of this half travels (its forward block of pack[0] has been consumed) ...
            for r in c.local_ranks:
                S = self.st[r]
                check(L.tlab_poisson_fft_z(S[key], -1, _ptr(pen(S, 2, h)), _ptr(pen(S, 1, h))), "fft_z")       # ... while dp^/dy is transformed
            back.append((wp, exchange(1, 1, h, nx_, False)))
        for i_pack, slab, out in ((0, 1, 0), (1, 3, 2)):             # p -> tmp1, dp/dy -> tmp3
            for wpair in back:
                wpair[i_pack].wait()
            for r in c.local_ranks:
                S = self.st[r]
                check(L.tlab_pencil_repack_blocks(_ptr(S["txc"][slab]), _ptr(S["pack"][i_pack]), self.nxh, ny, kmax, 2 * P, start, base, -1),
                      "tlab_pencil_repack_blocks")
                check(L.tlab_poisson_fft_x(S["poisson"], -1, _ptr(S["txc"][slab]), _ptr(S["txc"][out])), "fft_x")

    def _poisson_pencil(self):
        """OPR_Poisson_FourierXZ_Factorize on kx-pencils: forcing in tmp1, Neumann data in hb/ht; p -> tmp1, dp/dy -> tmp3."""
        if self.stages == 2:
            return self._poisson_pencil_staged()
        L = load()
        for r in self.comm.local_ranks:
            S = self.st[r]
            check(L.tlab_poisson_set_wall_planes(S["poisson"], _ptr(S["txc"][0]), _ptr(S["hb"]), _ptr(S["ht"])), "set_wall_planes")
            check(L.tlab_poisson_fft_x(S["poisson"], 1, _ptr(S["txc"][0]), _ptr(S["txc"][1])), "fft_x")          # p -> tmp2 (complex slab)
        self._pencil_forward(1).wait()
        for r in self.comm.local_ranks:
            S = self.st[r]
            b0, b1, b2 = S["pen"]
            check(L.tlab_poisson_fft_z(S["poisson"], 1, _ptr(b0), _ptr(b1)), "fft_z")
            if self.gy_elliptic is not None:
                check(L.tlab_poisson_direct_ode(S["poisson"], 3, _ptr(b1), _ptr(b1)), "direct_ode")                # p^ over f^ (BCS_NN)
            else:
                check(L.tlab_poisson_ode(S["poisson"], _ptr(b1), _ptr(b1), _ptr(b2)), "ode")                      # p^ over f^, dp^ in b2
            check(L.tlab_poisson_fft_z(S["poisson"], -1, _ptr(b1), _ptr(b0)), "fft_z")
        w0 = self._pencil_backward_start(0, 0)                                                                     # p travels ...
        if self.gy_elliptic is not None:     # one field on the way back; dp/dy = OPR_Partial_Y(p) on the slab (opr_elliptic.f90:447-449)
            w0.wait()
            self._pencil_backward_finish(0, 1)
            for r in self.comm.local_ranks:
                S = self.st[r]
                check(L.tlab_poisson_fft_x(S["poisson"], -1, _ptr(S["txc"][1]), _ptr(S["txc"][0])), "fft_x")     # p -> tmp1
                self._partial(2, self.g[1], self.nx, self.ny, self.kmax, S["txc"][0], S["txc"][2])                 # dp/dy -> tmp3
            return
        for r in self.comm.local_ranks:
            S = self.st[r]
            check(L.tlab_poisson_fft_z(S["poisson"], -1, _ptr(S["pen"][2]), _ptr(S["pen"][1])), "fft_z")          # ... while dp/dy is transformed
        w1 = self._pencil_backward_start(1, 1)
        w0.wait()
        self._pencil_backward_finish(0, 1)
        for r in self.comm.local_ranks:
            S = self.st[r]
            check(L.tlab_poisson_fft_x(S["poisson"], -1, _ptr(S["txc"][1]), _ptr(S["txc"][0])), "fft_x")         # p -> tmp1
        w1.wait()
        self._pencil_backward_finish(1, 3)
        for r in self.comm.local_ranks:
            S = self.st[r]
            check(L.tlab_poisson_fft_x(S["poisson"], -1, _ptr(S["txc"][3]), _ptr(S["txc"][2])), "fft_x")         # dp/dy -> tmp3

    def _rhs_halo(self, dte, tail=None):
        """Same terms as rhs.cpp / rhs_global_incompressible_1.f90:98-398; the z-terms are added last in every equation so that the
        neighbour messages travel while the x/y operators run (the reference's order differs in the third equation: rounding only).
        tail = (dte, kco, scale): fold the RK update into the last pass (TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT)."""
        L = load()
        nx, ny, kmax, n = self.nx, self.ny, self.kmax, self.n
        gx, gy = self.g[0], self.g[1]
        nu = self.visc
        ns = self.nscal
        T = lambda S, i: S["txc"][i]          # noqa: E731
        eqs = lambda S: [(S["q"][0], S["hq"][0], nu), (S["q"][1], S["hq"][1], nu), (S["q"][2], S["hq"][2], nu)] + \
            [(S["s"][i], S["hs"][i], self.visc / self.schmidt[i]) for i in range(ns)]            # noqa: E731

        def badd_all(d, g, S, overwrite=False):
            """hq, hs += Burgers_d of every transported field, four per launch (they share the advecting velocity q_d)."""
            E = eqs(S)
            for e0 in range(0, len(E), 4):
                grp = E[e0:e0 + 4]
                nf = len(grp)
                nus = (ctypes.c_double * nf)(*[float(kap) for _, _, kap in grp])
                sp = (c_vp * nf)(*[f.data_ptr() for f, _, _ in grp])
                rp = (c_vp * nf)(*[h.data_ptr() for _, h, _ in grp])
                check(L.tlab_opr_burgers_add_n(d, g._h, nx, ny, kmax, 0, nf, nus, sp, _ptr(S["q"][d - 1]), rp, _ptr(T(S, 6)), _ptr(T(S, 7)),
                                               int(overwrite)), "tlab_opr_burgers_add_n")

        def padd(d, g, S, u, ub, scale, res, acc):
            check(L.tlab_opr_partial_add(d, g._h, nx, ny, kmax, 0, _ptr(u), _ptr(ub) if ub is not None else None, float(scale), _ptr(res), int(acc),
                                         _ptr(T(S, 6)), _ptr(T(S, 7))), "tlab_opr_partial_add")

        # ---- diffusion + advection (:98-162) ----
        w = self._halo_start([("q", 0), ("q", 1), ("q", 2)] + [("s", i) for i in range(ns)])
        fresh, self._fresh = self._fresh, False        # start of a Runge-Kutta step: hq = hs = 0 (time.f90:212-216) -> the x-terms overwrite
        self._local(lambda r, S: badd_all(1, gx, S, fresh))
        w.wait()
        def zburgers_all(phase, S):
            E = eqs(S)
            for e0 in range(0, len(E), 4):
                grp = E[e0:e0 + 4]
                nf = len(grp)
                nus = (ctypes.c_double * nf)(*[float(kap) for _, _, kap in grp])
                sp = (c_vp * nf)(*[f.data_ptr() for f, _, _ in grp])
                rp = (c_vp * nf)(*[h.data_ptr() for _, h, _ in grp])
                o = 2 * e0 * self.npage * 8
                P = lambda t: c_vp(t.data_ptr() + o)                 # noqa: E731
                check(L.tlab_zslab_burgers_z_n(S["zplan"], phase, nx, ny, nf, nus, sp, _ptr(S["q"][2]) if phase == 2 else None,
                                               P(S["head"]), P(S["tail"]), P(S["tail_left"]), P(S["head_right"]), rp if phase == 2 else None, 1),
                      "tlab_zslab_burgers_z_n")
        self._local(lambda r, S: zburgers_all(1, S))
        w = self._msg_start(2 * (3 + ns))
        self._local(lambda r, S: badd_all(2, gy, S))
        w.wait()
        self._local(lambda r, S: zburgers_all(2, S))
        # ---- pressure forcing: div(hq + q/dte) (:188-260) ----
        idte = 1.0 / dte
        w = self._halo_start([("hq", 2)])                                   # w's halo planes are still valid
        self._local(lambda r, S: padd(2, gy, S, S["hq"][1], S["q"][1], idte, T(S, 0), 0))
        self._local(lambda r, S: padd(1, gx, S, S["hq"][0], S["q"][0], idte, T(S, 0), 1))
        w.wait()
        self._local(lambda r, S: self._zpartial(1, S, S["hq"][2], S["q"][2], idte, None, 0))
        w = self._msg_start(1)
        self._local(lambda r, S: check(L.tlab_pw_get_wall_planes(_ptr(S["hq"][1]), _ptr(S["hb"]), _ptr(S["ht"]), nx, ny, kmax), "walls"))
        w.wait()
        self._local(lambda r, S: self._zpartial(2, S, S["hq"][2], S["q"][2], idte, T(S, 0), 1))
        # ---- pressure (:284) and its gradient (:319-320) ----
        self._poisson_pencil()
        types = list(zip(self.flow_jmin, self.flow_jmax)) + list(zip(self.scal_jmin, self.scal_jmax))
        grad_final = tail is not None and all(t == DNS_BCS_DIRICHLET for pair in types[:3] for t in pair)
        # ---- hq -= grad p, boundary conditions (:348-398) [+ RK update] ----
        def finish(r, S):
            grads = [T(S, 1), T(S, 2), T(S, 3)] + [None] * ns
            fields = list(zip(S["q"] + S["s"], S["hq"] + S["hs"], grads, types))
            if grad_final:
                fields = [fields[1]] + fields[3:]          # v and the scalars; u, w are done
            neumann_vel = any(t == DNS_BCS_NEUMANN for pair in types[:3] for t in pair)
            if not grad_final and (neumann_vel or tail is None):
                check(L.tlab_pw_sub3(_ptr(S["hq"][0]), _ptr(S["hq"][1]), _ptr(S["hq"][2]), _ptr(T(S, 1)), _ptr(T(S, 2)), _ptr(T(S, 3)), n), "sub3")
                fields = [(q, h, None, t) for q, h, _, t in fields]
            for q, h, g, (tmin, tmax) in fields:
                ibc = (1 if tmin == DNS_BCS_NEUMANN else 0) + (2 if tmax == DNS_BCS_NEUMANN else 0)
                if ibc:       # needs the finished tendency (g is None here)
                    check(L.tlab_boundary_bcs_neumann_y(gy._h, ibc, nx, ny, kmax, _ptr(h), _ptr(S["hb"]), _ptr(S["ht"]), _ptr(T(S, 0))), "bcs_neumann_y")
                pb, pt = (_ptr(S["hb"]) if ibc & 1 else None), (_ptr(S["ht"]) if ibc & 2 else None)
                if tail is None:
                    check(L.tlab_pw_set_wall_planes(_ptr(h), pb, pt, nx, ny, kmax), "walls")
                else:
                    check(L.tlab_pw_final_update(_ptr(q), _ptr(h), _ptr(g) if g is not None else None, pb, pt, float(tail[0]), float(tail[1]),
                                                 int(tail[2]), nx, ny, kmax), "final_update")
        w = self._halo_start([("txc", 0)])
        if grad_final:   # u and w are finished by the gradient kernels themselves (no gradient array)
            self._local(lambda r, S: check(L.tlab_opr_gradient_final(1, gx._h, nx, ny, kmax, _ptr(T(S, 0)), _ptr(S["q"][0]), _ptr(S["hq"][0]), float(tail[0]),
                                                                     float(tail[1]), int(tail[2]), _ptr(T(S, 1))), "tlab_opr_gradient_final"))
        else:
            self._local(lambda r, S: padd(1, gx, S, T(S, 0), None, 0.0, T(S, 1), 0))
        w.wait()
        self._local(lambda r, S: self._zpartial(1, S, T(S, 0), None, 0.0, None, 0))
        w = self._msg_start(1)
        # v and the scalars do not wait for dp/dz: their update runs while the interface values travel (not with Neumann scalars, whose
        # boundary routine takes tmp1 = p as scratch)
        early_finish = grad_final and all(t == DNS_BCS_DIRICHLET for pair in types[3:] for t in pair)
        if early_finish:
            self._local(lambda r, S: finish(r, S))
        w.wait()
        if grad_final:
            self._local(lambda r, S: check(L.tlab_zslab_gradient_final_z(S["zplan"], nx, ny, _ptr(T(S, 0)), _ptr(S["tail_left"]), _ptr(S["head_right"]),
                                                                         _ptr(S["q"][2]), _ptr(S["hq"][2]), float(tail[0]), float(tail[1]), int(tail[2])),
                                           "tlab_zslab_gradient_final_z"))
        else:
            self._local(lambda r, S: self._zpartial(2, S, T(S, 0), None, 0.0, T(S, 3), 0))
        if not early_finish:
            self._local(finish)

    # ---- the RHS ---------------------------------------------------------------------------------------------------------
    def RHS_GLOBAL_INCOMPRESSIBLE_1(self, dte):
        _use_torch_stream()
        if self.zmode == "halo":
            return self._rhs_halo(dte)
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

    # ---- per-iteration monitors on slabs: local device reductions + one MPI_MAX / MPI_MIN of two scalars ----
    def _dns_handle(self, r):
        """tlab_dns handle of the local box (nx, ny, kmax) with the GLOBAL z plan and the slab's first plane (ims_offset_k)."""
        S = self.st[r]
        if "dns" not in S:
            h = c_vp(0)
            sc = np.ascontiguousarray(self.schmidt if self.nscal else [1.0], dtype=np.float64)
            check(load().tlab_dns_create(ctypes.byref(h), self.g[0]._h, self.g[1]._h, self.g[2]._h, S["poisson"], self.nx, self.ny, self.kmax,
                                         self.nscal, self.visc, sc.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), "tlab_dns_create")
            check(load().tlab_dns_set_slab(h, r * self.kmax), "tlab_dns_set_slab")
            S["dns"] = h
        return S["dns"]

    def TIME_COURANT(self, cfla, cfld):
        """tools/dns/time.f90:365-548 with the MPI_MAX of :522.  Returns ((pmax1, pmax2), dtime), the same on every rank."""
        _use_torch_stream()
        loc = {}
        for r in self.comm.local_ranks:
            q = (c_vp * 3)(*[t.data_ptr() for t in self.st[r]["q"]])
            pmax = (ctypes.c_double * 2)()
            check(load().tlab_time_courant(self._dns_handle(r), q, float(cfla), float(cfld), pmax, None), "tlab_time_courant")
            loc[r] = [pmax[0], pmax[1]]
        p1, p2 = self.comm.all_reduce(loc, "max")
        dtc = cfla / p1 if p1 > 0.0 else 1.0e300
        dtd = cfld / p2 if p2 > 0.0 else 1.0e300
        return (p1, p2), (min(dtc, dtd) if cfla > 0.0 else 0.0)

    def dilatation_bounds(self):
        """DNS_BOUNDS_CONTROL (dns_local.f90:157-187): (DilMin, DilMax) = extremes of div(q) (FI_INVARIANT_P = -div, fi_vectorcalculus.f90:111-141);
        the z-derivative takes the slab route of the RHS (halo planes + interface values, or the K-transposes)."""
        _use_torch_stream()
        L = load()
        nx, ny, kmax, n = self.nx, self.ny, self.kmax, self.n
        T = lambda S, i: S["txc"][i]        # noqa: E731
        halo = self.zmode == "halo"
        w = self._halo_start([("q", 2)]) if halo else None
        self._local(lambda r, S: self._partial(1, self.g[0], nx, ny, kmax, S["q"][0], T(S, 0)))
        self._local(lambda r, S: self._partial(2, self.g[1], nx, ny, kmax, S["q"][1], T(S, 1)))
        self._local(lambda r, S: T(S, 0)[:n].add_(T(S, 1)[:n]))
        if halo:
            w.wait()
            self._local(lambda r, S: self._zpartial(1, S, S["q"][2], None, 0.0, None, 0))
            self._msg_start(1).wait()
            self._local(lambda r, S: self._zpartial(2, S, S["q"][2], None, 0.0, T(S, 0), 1))
        else:
            self._local(lambda r, S: T(S, 3)[:n].copy_(S["q"][2]))
            self.partial_z(3, 2)
            self._local(lambda r, S: T(S, 0)[:n].add_(T(S, 2)[:n]))
        loc_mn, loc_mx = {}, {}
        for r in self.comm.local_ranks:
            S = self.st[r]
            mn, mx = ctypes.c_double(0.0), ctypes.c_double(0.0)
            check(L.tlab_minmax(self._dns_handle(r), _ptr(T(S, 0)), nx, ny, kmax, ctypes.byref(mn), ctypes.byref(mx)), "tlab_minmax")
            loc_mn[r], loc_mx[r] = [mn.value], [mx.value]
        return self.comm.all_reduce(loc_mn, "min")[0], self.comm.all_reduce(loc_mx, "max")[0]

    def TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(self, dte, kco=1.0, scale_tendencies=False):
        if self.zmode == "halo":
            _use_torch_stream()
            return self._rhs_halo(dte, tail=(dte, kco, scale_tendencies))
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
            if self.zmode == "halo":
                self._fresh = True
            else:
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
                if "poisson_b" in S:
                    load().tlab_poisson_plan_destroy(S["poisson_b"])
                if "zplan" in S:
                    load().tlab_zslab_plan_destroy(S["zplan"])
                if "dns" in S:
                    load().tlab_dns_destroy(S["dns"])
        except Exception:
            pass
