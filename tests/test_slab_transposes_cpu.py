"""CPU tests of the z-slab communication layer (tlab_amd/parallel.py): the K-transposes are pure index work and must be
bit-exact.  (1) single process, all ranks simulated (LoopbackComm): the forward layout equals the serial layout of the global
array restricted to the pencil, and backward(forward(a)) == a  (the reference's own check: operators/opr_check.f90:71-91,
valid/mpi/vmpi_transpose.f90:152-199).  (2) two real processes over gloo produce the same blocks as the simulation."""
import os
import socket
import numpy as np
import pytest
import torch

from tlab_amd.parallel import LoopbackComm, DistComm, trp_k_forward, trp_k_backward, trp_i_forward, trp_i_backward, pencil_stage_layout


def global_field(nx, ny, nz, width):
    # value encodes its own global index so that any misplaced element is detected
    return torch.arange(nx * ny * nz * width, dtype=torch.float64) * 1.0 + 0.25


@pytest.mark.parametrize("P,nx,ny,nz,width", [(2, 8, 6, 4, 1), (4, 8, 6, 8, 1), (2, 5, 4, 6, 2), (8, 16, 8, 16, 1), (3, 6, 3, 9, 2)])
def test_k_transposes_loopback(P, nx, ny, nz, width):
    comm = LoopbackComm(P)
    kmax, npage = nz // P, nx * ny
    nl = npage // P
    g = global_field(nx, ny, nz, width)
    a = {r: g[r * npage * kmax * width:(r + 1) * npage * kmax * width].clone() for r in range(P)}
    b = trp_k_forward(comm, a, npage, kmax, width)
    G = g.view(nz, npage, width)
    for r in range(P):
        # rank r owns in-plane indices [r*nl, (r+1)*nl) for ALL z: b(l, z) lines-fastest
        expect = G[:, r * nl:(r + 1) * nl, :].reshape(-1)
        assert torch.equal(b[r], expect), r
    back = trp_k_backward(comm, b, npage, kmax, width)
    for r in range(P):
        assert torch.equal(back[r], a[r]), r


def _worker(rank, world, port, nx, ny, nz, width, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = DistComm()
        kmax, npage = nz // world, nx * ny
        g = global_field(nx, ny, nz, width)
        a = {rank: g[rank * npage * kmax * width:(rank + 1) * npage * kmax * width].clone()}
        b = trp_k_forward(comm, a, npage, kmax, width)
        back = trp_k_backward(comm, b, npage, kmax, width)
        nl = npage // world
        expect = g.view(nz, npage, width)[:, rank * nl:(rank + 1) * nl, :].reshape(-1)
        q.put((rank, bool(torch.equal(b[rank], expect)), bool(torch.equal(back[rank], a[rank]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("width", [1, 2])
def test_k_transposes_two_processes_gloo(width):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, 6, 4, width, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res


# ---------------------------------------------------------------------------------------------------------------------------------
# primitives of the transpose-free algorithm: ring exchange with the two neighbours, uneven all-to-all of the kx-pencils
# ---------------------------------------------------------------------------------------------------------------------------------
def _ring_payload(P):
    """Per rank: two tensors for the left and two for the right neighbour, values encode (sender, direction, index)."""
    tl = {r: [torch.full((5,), 100.0 * r + 1.0), torch.full((3,), 100.0 * r + 2.0)] for r in range(P)}
    tr = {r: [torch.full((5,), 100.0 * r + 3.0), torch.full((3,), 100.0 * r + 4.0)] for r in range(P)}
    return tl, tr


def _ring_expect(P, r):
    right, left = (r + 1) % P, (r - 1) % P
    return ([torch.full((5,), 100.0 * right + 1.0), torch.full((3,), 100.0 * right + 2.0)],      # what my right neighbour sent left
            [torch.full((5,), 100.0 * left + 3.0), torch.full((3,), 100.0 * left + 4.0)])        # what my left neighbour sent right


@pytest.mark.parametrize("P", [2, 3, 8])
def test_neighbor_exchange_loopback(P):
    comm = LoopbackComm(P)
    tl, tr = _ring_payload(P)
    fr = {r: [torch.zeros(5), torch.zeros(3)] for r in range(P)}
    fl = {r: [torch.zeros(5), torch.zeros(3)] for r in range(P)}
    comm.neighbor_exchange(tl, tr, fr, fl).wait()
    for r in range(P):
        er, el = _ring_expect(P, r)
        assert all(torch.equal(a, b) for a, b in zip(fr[r], er)) and all(torch.equal(a, b) for a, b in zip(fl[r], el)), r


def _pencil_case(P, nxh, ny, kmax):
    base, rem = divmod(nxh, P)
    nxl = [base + (1 if r < rem else 0) for r in range(P)]
    ioff = [r * base + min(r, rem) for r in range(P)]
    nz = kmax * P
    g = torch.arange(nz * ny * nxh, dtype=torch.float64).view(nz, ny, nxh)       # global spectral array, value = its own index
    return nxl, ioff, g


def _pencil_send(P, r, nxl, ioff, g, ny, kmax):
    a = g[r * kmax:(r + 1) * kmax]                                               # my slab (kmax, ny, nxh)
    blocks = [a[:, :, ioff[p]:ioff[p] + nxl[p]].reshape(-1) for p in range(P)]
    return torch.cat(blocks), [b.numel() for b in blocks]


@pytest.mark.parametrize("P,nxh,ny,kmax", [(2, 9, 3, 2), (3, 17, 2, 2), (8, 33, 2, 1)])
def test_pencil_all_to_all_loopback(P, nxh, ny, kmax):
    comm = LoopbackComm(P)
    nxl, ioff, g = _pencil_case(P, nxh, ny, kmax)
    send, scnt, recv, rcnt = {}, {}, {}, {}
    for r in range(P):
        send[r], scnt[r] = _pencil_send(P, r, nxl, ioff, g, ny, kmax)
        recv[r], rcnt[r] = torch.zeros(kmax * P * ny * nxl[r], dtype=torch.float64), [kmax * ny * nxl[r]] * P
    comm.all_to_all_v(send, scnt, recv, rcnt).wait()
    for r in range(P):      # the pencil of rank r: all z, its kx range, with NO unpacking on the receive side
        assert torch.equal(recv[r].view(kmax * P, ny, nxl[r]), g[:, :, ioff[r]:ioff[r] + nxl[r]]), r


def _worker_ring(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = DistComm()
        tl, tr = _ring_payload(world)
        fr = {rank: [torch.zeros(5), torch.zeros(3)]}
        fl = {rank: [torch.zeros(5), torch.zeros(3)]}
        comm.neighbor_exchange({rank: tl[rank]}, {rank: tr[rank]}, fr, fl).wait()
        er, el = _ring_expect(world, rank)
        ok_ring = all(torch.equal(a, b) for a, b in zip(fr[rank], er)) and all(torch.equal(a, b) for a, b in zip(fl[rank], el))
        nxh, ny, kmax = 9, 3, 2
        nxl, ioff, g = _pencil_case(world, nxh, ny, kmax)
        s, sc = _pencil_send(world, rank, nxl, ioff, g, ny, kmax)
        rv = {rank: torch.zeros(kmax * world * ny * nxl[rank], dtype=torch.float64)}
        comm.all_to_all_v({rank: s}, {rank: sc}, rv, {rank: [kmax * ny * nxl[rank]] * world}).wait()
        ok_pen = torch.equal(rv[rank].view(kmax * world, ny, nxl[rank]), g[:, :, ioff[rank]:ioff[rank] + nxl[rank]])
        q.put((rank, bool(ok_ring), bool(ok_pen)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_and_pencil_exchanges_real_processes_gloo(world):
    """2 ranks: both neighbours are the same peer (the posting order in DistComm.neighbor_exchange must pair the messages);
    3 ranks: distinct neighbours; uneven kx ranges (9 = 5 + 4 = 3 + 3 + 3)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ring, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] and r[2] for r in res), res


def _staged_pack(P, r, ioff, nxl, g, ny, kmax):
    """What tlab_pencil_repack_blocks writes for rank r's slab (one value per element here): blocks [kmax][ny][width] at base[b]."""
    start, base, split, nxa, nxb = pencil_stage_layout(ioff, nxl, ny, kmax)
    a = g[r * kmax:(r + 1) * kmax]
    buf = torch.full((a.numel(),), -1.0, dtype=torch.float64)
    ends = start[1:] + [g.shape[2]]
    for b in range(2 * P):
        blk = a[:, :, start[b]:ends[b]].reshape(-1)
        buf[base[b]:base[b] + blk.numel()] = blk
    return buf, split, nxa, nxb


def _staged_exchange(comm, ranks, P, ioff, nxl, g, ny, kmax):
    """Both halves forward through views of ONE pack buffer per rank (as SlabDns._poisson_pencil_staged posts them); returns {rank: (penA, penB)}."""
    packs, pens = {}, {}
    for r in ranks:
        packs[r], split, nxa, nxb = _staged_pack(P, r, ioff, nxl, g, ny, kmax)
        assert not bool((packs[r] < 0).any())                      # the blocks tile the buffer
        pens[r] = (torch.zeros(kmax * P * ny * nxa[r], dtype=torch.float64), torch.zeros(kmax * P * ny * nxb[r], dtype=torch.float64))
    works = []
    for h, nxh_ in ((0, nxa), (1, nxb)):
        send = {r: (packs[r][:split] if h == 0 else packs[r][split:]) for r in ranks}
        scnt = {r: [nxh_[p] * ny * kmax for p in range(P)] for r in ranks}
        recv = {r: pens[r][h] for r in ranks}
        rcnt = {r: [nxh_[r] * ny * kmax] * P for r in ranks}
        works.append(comm.all_to_all_v(send, scnt, recv, rcnt))
    for w in works:
        w.wait()
    return pens, nxa, nxb


@pytest.mark.parametrize("P,nxh,ny,kmax", [(2, 9, 3, 2), (3, 17, 2, 2), (8, 33, 2, 1)])
def test_staged_pencil_exchange_loopback(P, nxh, ny, kmax):
    """Two-stage pencil exchange: halves A and B of every kx range arrive as (nz, ny, half width) with no unpacking, uneven ranges included."""
    nxl, ioff, g = _pencil_case(P, nxh, ny, kmax)
    pens, nxa, nxb = _staged_exchange(LoopbackComm(P), range(P), P, ioff, nxl, g, ny, kmax)
    for r in range(P):
        assert torch.equal(pens[r][0].view(kmax * P, ny, nxa[r]), g[:, :, ioff[r]:ioff[r] + nxa[r]]), r
        assert torch.equal(pens[r][1].view(kmax * P, ny, nxb[r]), g[:, :, ioff[r] + nxa[r]:ioff[r] + nxl[r]]), r


def _worker_staged(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nxh, ny, kmax = 9, 3, 2
        nxl, ioff, g = _pencil_case(world, nxh, ny, kmax)
        pens, nxa, nxb = _staged_exchange(DistComm(), [rank], world, ioff, nxl, g, ny, kmax)
        ok = torch.equal(pens[rank][0].view(kmax * world, ny, nxa[rank]), g[:, :, ioff[rank]:ioff[rank] + nxa[rank]]) and \
            torch.equal(pens[rank][1].view(kmax * world, ny, nxb[rank]), g[:, :, ioff[rank] + nxa[rank]:ioff[rank] + nxl[rank]])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_staged_pencil_exchange_two_processes_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_staged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res), res


def _packed_by_map(P, r, ioff, nxl, g, ny, kmax):
    """What the packed x-transform of the native slab driver writes (tlab_poisson_fft_x_packed: element (line, kx) -> off[kx] + line * width[kx],
    the map computed by the library on the host), for the staged block map of rank r's slab."""
    import ctypes
    from tlab_amd.lib import load, check
    start, base, split, nxa, nxb = pencil_stage_layout(ioff, nxl, ny, kmax)
    nxh = g.shape[2]
    st = (ctypes.c_int * len(start))(*start)
    bs = (ctypes.c_longlong * len(base))(*base)
    off = (ctypes.c_longlong * nxh)()
    wid = (ctypes.c_int * nxh)()
    check(load().tlab_debug_pack_map(nxh, len(start), st, bs, off, wid), "tlab_debug_pack_map")
    a = g[r * kmax:(r + 1) * kmax].reshape(kmax * ny, nxh)          # lines = (k, j), kx fastest
    buf = torch.full((a.numel(),), -1.0, dtype=torch.float64)
    for kx in range(nxh):
        buf[off[kx] + torch.arange(kmax * ny) * wid[kx]] = a[:, kx]
    return buf


@pytest.mark.parametrize("P,nxh,ny,kmax", [(2, 9, 3, 2), (3, 17, 2, 2), (8, 33, 2, 1), (1, 5, 2, 3)])
def test_packed_transform_layout_is_the_repack_layout(P, nxh, ny, kmax):
    """The repack pass folded into the x-transforms: writing element (line, kx) at off[kx] + line * width[kx] fills the pack buffer exactly as
    tlab_pencil_repack_blocks does (no gaps, no overlaps), for even and uneven kx ranges and both halves of the staged exchange."""
    nxl, ioff, g = _pencil_case(P, nxh, ny, kmax)
    for r in range(P):
        ref, split, nxa, nxb = _staged_pack(P, r, ioff, nxl, g, ny, kmax)
        got = _packed_by_map(P, r, ioff, nxl, g, ny, kmax)
        assert torch.equal(got, ref), r


def _worker_packed(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nxh, ny, kmax = 9, 3, 2
        nxl, ioff, g = _pencil_case(world, nxh, ny, kmax)
        start, base, split, nxa, nxb = pencil_stage_layout(ioff, nxl, ny, kmax)
        pack = _packed_by_map(world, rank, ioff, nxl, g, ny, kmax)
        comm = DistComm()
        pens = (torch.zeros(kmax * world * ny * nxa[rank], dtype=torch.float64), torch.zeros(kmax * world * ny * nxb[rank], dtype=torch.float64))
        works = []
        for h, w in ((0, nxa), (1, nxb)):
            works.append(comm.all_to_all_v({rank: pack[:split] if h == 0 else pack[split:]}, {rank: [w[p] * ny * kmax for p in range(world)]},
                                           {rank: pens[h]}, {rank: [w[rank] * ny * kmax] * world}))
        for wk in works:
            wk.wait()
        ok = torch.equal(pens[0].view(kmax * world, ny, nxa[rank]), g[:, :, ioff[rank]:ioff[rank] + nxa[rank]]) and \
            torch.equal(pens[1].view(kmax * world, ny, nxb[rank]), g[:, :, ioff[rank] + nxa[rank]:ioff[rank] + nxl[rank]])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_packed_transform_layout_two_processes_gloo():
    """World size 2 over gloo: pack buffers filled through the library's map travel as the two halves of the staged exchange and arrive as the
    kx-pencils (nz, ny, half width) with no unpacking."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res), res


# ---------------------------------------------------------------------------------------------------------------------------------
# I-transposes (x pencils): TLabMPI_Trp_ExecI_*, tlab_mpi_transpose.f90:205-286
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,imax,ny,nz,width", [(2, 4, 3, 2, 1), (4, 8, 2, 6, 1), (2, 5, 2, 4, 2), (8, 2, 4, 4, 1)])
def test_i_transposes_loopback(P, imax, ny, nz, width):
    comm = LoopbackComm(P)
    nxg, npage = imax * P, ny * nz
    nl = npage // P
    g = torch.arange(nxg * npage * width, dtype=torch.float64).view(npage, nxg, width) + 0.5        # global (line, x_global)
    a = {r: g[:, r * imax:(r + 1) * imax, :].reshape(-1).clone() for r in range(P)}                 # rank r owns the x range [r*imax, (r+1)*imax)
    b = trp_i_forward(comm, a, imax, npage, width)
    for r in range(P):       # rank r now owns the complete x-lines [r*nl, (r+1)*nl)
        assert torch.equal(b[r], g[r * nl:(r + 1) * nl].reshape(-1)), r
    back = trp_i_backward(comm, b, imax, npage, width)
    for r in range(P):
        assert torch.equal(back[r], a[r]), r


def _worker_i(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = DistComm()
        imax, ny, nz, width = 4, 3, 2, 1
        nxg, npage = imax * world, ny * nz
        nl = npage // world
        g = torch.arange(nxg * npage * width, dtype=torch.float64).view(npage, nxg, width) + 0.5
        a = {rank: g[:, rank * imax:(rank + 1) * imax, :].reshape(-1).clone()}
        b = trp_i_forward(comm, a, imax, npage, width)
        back = trp_i_backward(comm, b, imax, npage, width)
        q.put((rank, bool(torch.equal(b[rank], g[rank * nl:(rank + 1) * nl].reshape(-1))), bool(torch.equal(back[rank], a[rank]))))
    finally:
        dist.destroy_process_group()


def test_i_transposes_two_processes_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_i, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] and r[2] for r in res), res


def _worker_allreduce(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = DistComm()
        mx = comm.all_reduce({rank: [1.0 + rank, -3.0 * rank]}, "max")
        mn = comm.all_reduce({rank: [1.0 + rank, -3.0 * rank]}, "min")
        q.put((rank, mx, mn))
    finally:
        dist.destroy_process_group()


def test_all_reduce_two_processes_gloo():
    """MPI_ALLREDUCE(MPI_MAX) of TIME_COURANT (time.f90:522) and the MINMAX pair of DNS_BOUNDS_CONTROL over the z communicator."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_allreduce, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for _, mx, mn in res:
        assert mx == [2.0, 0.0] and mn == [1.0, -3.0]
    assert LoopbackComm(3).all_reduce({0: [1.0], 1: [5.0], 2: [-2.0]}, "max") == [5.0]
