"""CPU tests of the z-slab communication layer (tlab_amd/parallel.py): the K-transposes are pure index work and must be
bit-exact.  (1) single process, all ranks simulated (LoopbackComm): the forward layout equals the serial layout of the global
array restricted to the pencil, and backward(forward(a)) == a  (the reference's own check: operators/opr_check.f90:71-91,
valid/mpi/vmpi_transpose.f90:152-199).  (2) two real processes over gloo produce the same blocks as the simulation."""
import os
import socket
import numpy as np
import pytest
import torch

from tlab_amd.parallel import LoopbackComm, DistComm, trp_k_forward, trp_k_backward


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
