#!/usr/bin/env python3
"""Check of the native transposition layer (libtlab_amd_comm.so, include/tlab_amd_comm.h) -- no torch in this process.

  1. loopback: P ranks simulated in this one process with the two halves tlab_trp_pack / tlab_trp_unpack around a host-side "wire"
     (block q of rank r's receive buffer = block r of rank q's send buffer): I and K transpositions, real and complex, forward and backward,
     bit-exact against the closed form (rank r's pencil of the global array, base/tlab_mpi_transpose.f90:232-256, :301-325).
  2. RCCL: tlab_comm_init over the ranks this script is started as (RANK / WORLD_SIZE, identifier through --idfile; default: one rank),
     tlab_trp_exec / tlab_trp_start + tlab_trp_wait and tlab_comm_allreduce_max through the communicator; with several ranks (one GPU each)
     every rank checks its pencil of a global array all ranks can compute.
Prints "native comm ok" on success."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tlab_amd.lib import load as load_core, check as check_core, c_vp  # noqa: E402
from tlab_amd import comm as C  # noqa: E402


class Dev:
    """device array of n doubles through tlab_malloc"""

    def __init__(self, n):
        self.n = int(n)
        self.p = c_vp(0)
        check_core(load_core().tlab_malloc(ctypes.byref(self.p), max(self.n, 1) * 8), "tlab_malloc")

    @property
    def ptr(self):
        return self.p.value

    def put(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        assert a.size == self.n
        check_core(load_core().tlab_memcpy_h2d(self.p, a.ctypes.data_as(c_vp), a.nbytes), "h2d")

    def get(self):
        a = np.empty(self.n)
        check_core(load_core().tlab_memcpy_d2h(a.ctypes.data_as(c_vp), self.p, a.nbytes), "d2h")
        return a

    def free(self):
        load_core().tlab_free(self.p)


def global_array(dir, nmax, npage, P, e, seed):
    """G in numpy order (slow ... fast): I: (npage, imax*P, e); K: (kmax*P, npage, e)"""
    rng = np.random.default_rng(seed)
    shape = (npage, nmax * P, e) if dir == 1 else (nmax * P, npage, e)
    return rng.uniform(-1, 1, shape)


def local_a(G, dir, nmax, r):
    return G[:, r * nmax:(r + 1) * nmax, :] if dir == 1 else G[r * nmax:(r + 1) * nmax, :, :]


def local_b(G, dir, nlines, r):
    return G[r * nlines:(r + 1) * nlines, :, :] if dir == 1 else G[:, r * nlines:(r + 1) * nlines, :]


def f32(a):
    """what TransposeType = single leaves of a field: every value rounded to single precision (tlab_mpi_transpose.f90:362-371)"""
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def loopback(dir, nmax, npage, P, e, single=False):
    G = global_array(dir, nmax, npage, P, e, 7 * dir + P + e + nmax)
    nl = npage // P
    plans = [C.TrpPlan(None, dir, nmax, npage, e, r, P) for r in range(P)]
    n = plans[0].info(4)
    blk = plans[0].info(3)
    assert n == nmax * npage * e and blk * P == n and plans[0].info(0) == nl
    if single:      # fp32 wire: the buffers hold npro blocks of floats; result = the transposition of the fp32-rounded field, bit for bit
        for p in plans:
            p.set_wire(True)
            assert p.info(5) == 4
        src, snd, rcv, dst = ([Dev(n) for _ in range(P)] for _ in range(4))
        for fwd in (1, 0):
            for r in range(P):
                src[r].put(local_a(G, dir, nmax, r) if fwd else local_b(G, dir, nl, r))
                plans[r].pack(fwd, src[r].ptr, snd[r].ptr)
            wire = [snd[r].get().view(np.float32)[:n].reshape(P, blk) for r in range(P)]
            for r in range(P):
                w = np.zeros(n, dtype=np.float64)
                w.view(np.float32)[:n] = np.stack([wire[q][r] for q in range(P)]).reshape(-1)
                rcv[r].put(w)
                plans[r].unpack(fwd, rcv[r].ptr, dst[r].ptr)
                want = f32(local_b(G, dir, nl, r) if fwd else local_a(G, dir, nmax, r))
                assert np.array_equal(dst[r].get(), np.ascontiguousarray(want).reshape(-1)), ("loopback fp32 wire", dir, nmax, npage, P, fwd, r)
        for d in src + snd + rcv + dst:
            d.free()
        for p in plans:
            p.close()
        return
    src, snd, rcv, dst = ([Dev(n) for _ in range(P)] for _ in range(4))
    for fwd in (1, 0):
        for r in range(P):
            src[r].put(local_a(G, dir, nmax, r) if fwd else local_b(G, dir, nl, r))
            plans[r].pack(fwd, src[r].ptr, snd[r].ptr)
        wire = [snd[r].get().reshape(P, blk) for r in range(P)]
        for r in range(P):
            rcv[r].put(np.stack([wire[q][r] for q in range(P)]))
            plans[r].unpack(fwd, rcv[r].ptr, dst[r].ptr)
            want = local_b(G, dir, nl, r) if fwd else local_a(G, dir, nmax, r)
            got = dst[r].get()
            assert np.array_equal(got, np.ascontiguousarray(want).reshape(-1)), ("loopback", dir, nmax, npage, P, e, fwd, r)
    for d in src + snd + rcv + dst:
        d.free()
    for p in plans:
        p.close()


def rccl(args):
    L = load_core()
    rank, world = args.rank, args.nranks
    if world > 1:
        if rank == 0:
            ident = C.unique_id()
            with open(args.idfile + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(args.idfile + ".tmp", args.idfile)
        else:
            t0 = time.time()
            while not os.path.exists(args.idfile):
                if time.time() - t0 > 120:
                    sys.exit("rank %d: no identifier file" % rank)
                time.sleep(0.05)
            ident = open(args.idfile, "rb").read()
    else:
        ident = C.unique_id()
    comm = C.NativeComm(ident, world, rank, args.npro_i, world // args.npro_i)
    assert comm.info(0) == rank and comm.info(2) == rank % args.npro_i and comm.info(4) == rank // args.npro_i
    pi, pk, ni, nk = comm.info(2), comm.info(4), comm.info(3), comm.info(5)
    # all-reduce (TIME_COURANT's MPI_MAX)
    v = Dev(2)
    v.put(np.array([1.0 + rank, -3.0 - rank]))
    comm.allreduce_max(v.ptr, 2)
    L.tlab_sync()
    assert np.array_equal(v.get(), np.array([float(world), -3.0])), v.get()
    for dir, P, r in ((1, ni, pi), (3, nk, pk)):
        for e in (1, 2):
            nmax, npage = (48, 8 * 6 * P) if dir == 1 else (12, 10 * 8 * P)
            G = global_array(dir, nmax, npage, P, e, 100 + dir + e + (pk if dir == 1 else pi))      # the same array on the ranks of one communicator
            nl = npage // P
            plan = C.TrpPlan(comm, dir, nmax, npage, e)
            assert plan.info(1) == P and plan.info(2) == r
            n = plan.info(4)
            a, b, a2 = Dev(n), Dev(n), Dev(n)
            a.put(local_a(G, dir, nmax, r))
            plan.exec(1, a.ptr, b.ptr)
            L.tlab_sync()
            assert np.array_equal(b.get(), np.ascontiguousarray(local_b(G, dir, nl, r)).reshape(-1)), ("rccl forward", dir, e, rank)
            plan.start(0, b.ptr, a2.ptr)          # split form: other work may be enqueued here
            plan.wait()
            L.tlab_sync()
            assert np.array_equal(a2.get(), np.ascontiguousarray(local_a(G, dir, nmax, r)).reshape(-1)), ("rccl backward", dir, e, rank)
            if e == 1:      # the same through the fp32 wire: forward = T(fp32(a)); backward of that result returns it unchanged (already fp32 values)
                plan.set_wire(True)
                plan.exec(1, a.ptr, b.ptr)
                L.tlab_sync()
                assert np.array_equal(b.get(), np.ascontiguousarray(f32(local_b(G, dir, nl, r))).reshape(-1)), ("rccl fp32 forward", dir, rank)
                plan.start(0, b.ptr, a2.ptr)
                plan.wait()
                L.tlab_sync()
                assert np.array_equal(a2.get(), np.ascontiguousarray(f32(local_a(G, dir, nmax, r))).reshape(-1)), ("rccl fp32 backward", dir, rank)
            else:
                try:
                    plan.set_wire(True)
                    raise AssertionError("complex plans must refuse the fp32 wire")
                except C.TlabError:
                    pass
            plan.close()
            for d in (a, b, a2):
                d.free()
    comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nranks", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--rank", type=int, default=int(os.environ.get("RANK", "0")))
    ap.add_argument("--npro-i", type=int, default=1)
    ap.add_argument("--idfile", default="/tmp/tlab_amd_comm.id")
    ap.add_argument("--device", type=int, default=None)
    args = ap.parse_args()
    dev = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", args.rank))
    check_core(load_core().tlab_init(dev), "tlab_init")
    if args.rank == 0:
        for dir, nmax, npage in ((1, 24, 60), (1, 17, 24), (3, 10, 120), (3, 7, 24)):
            for P in (1, 2, 3, 4):
                if npage % P:
                    continue
                for e in (1, 2):
                    loopback(dir, nmax, npage, P, e)
        loopback(3, 8, 8 * 64, 8, 2)
        loopback(1, 64, 8 * 16, 8, 1)
        for dir, nmax, npage, P in ((1, 24, 60, 3), (1, 17, 24, 2), (3, 10, 120, 4), (3, 7, 24, 1), (3, 8, 8 * 64, 8), (1, 64, 8 * 16, 8)):
            loopback(dir, nmax, npage, P, 1, single=True)
    rccl(args)
    print("native comm ok (rank %d of %d)" % (args.rank, args.nranks))


if __name__ == "__main__":
    main()
