"""CPU tests of the communication layer of the x/z pencil driver (tlab_amd/pencil.py): the cartesian rank layout of TLabMPI_Initialize
(base/tlab_mpi_procs.f90:76-94), I-transpositions inside the x communicators and K-transpositions inside the z communicators, all keyed by
world rank.  The property the decomposed Poisson solver rests on is checked explicitly: after the I-transposition of a block field, world
rank r holds complete x lines of the z planes [r kmax/npro_i, (r + 1) kmax/npro_i) -- a 1 x (npro_i npro_k) slab decomposition in rank order.
(1) all ranks simulated in one process; (2) four real processes (2 x 2) over gloo."""
import os
import socket
import pytest
import torch

from tlab_amd.parallel import trp_k_forward, trp_k_backward, trp_i_forward, trp_i_backward
from tlab_amd.pencil import cart_groups, loopback_comms, dist_comms


def blocks(g, npi, npk):
    """{world rank: flat block (imax, ny, kmax)} of a global field g[nz][ny][nx]"""
    nz, ny, nx = g.shape
    imax, kmax = nx // npi, nz // npk
    return {pk * npi + pi: g[pk * kmax:(pk + 1) * kmax, :, pi * imax:(pi + 1) * imax].contiguous().reshape(-1) for pk in range(npk) for pi in range(npi)}


def check_rank(r, npi, npk, g, b_i, b_k, back_i, back_k, a):
    nz, ny, nx = g.shape
    imax, kmax = nx // npi, nz // npk
    kmax2 = kmax // npi
    pi, pk = r % npi, r // npi
    ok = torch.equal(b_i, g[r * kmax2:(r + 1) * kmax2].reshape(-1))                    # complete x lines of a z-slab in world-rank order
    nlz = imax * ny // npk                                                               # K: in-plane indices [pk nlz, (pk+1) nlz) of my x block, all z
    col = g[:, :, pi * imax:(pi + 1) * imax].reshape(nz, ny * imax)[:, pk * nlz:(pk + 1) * nlz]
    ok = ok and torch.equal(b_k, col.reshape(-1))
    return bool(ok and torch.equal(back_i, a) and torch.equal(back_k, a))


@pytest.mark.parametrize("npi,npk,nx,ny,nz", [(2, 2, 8, 3, 8), (2, 4, 8, 4, 16), (4, 2, 16, 2, 8), (1, 2, 4, 2, 4), (2, 1, 4, 3, 4), (8, 1, 16, 2, 8)])
def test_cartesian_transposes_loopback(npi, npk, nx, ny, nz):
    gx, gz = cart_groups(npi, npk)
    assert all(len(g) == npi for g in gx) and all(len(g) == npk for g in gz)
    assert sorted(r for g in gx for r in g) == list(range(npi * npk)) == sorted(r for g in gz for r in g)
    _, cx, cz = loopback_comms(npi, npk)
    g = torch.arange(nx * ny * nz, dtype=torch.float64).view(nz, ny, nx) + 0.5
    a = blocks(g, npi, npk)
    imax, kmax = nx // npi, nz // npk
    bi = trp_i_forward(cx, a, imax, ny * kmax)
    bk = trp_k_forward(cz, a, imax * ny, kmax)
    ri = trp_i_backward(cx, bi, imax, ny * kmax)
    rk = trp_k_backward(cz, bk, imax * ny, kmax)
    for r in range(npi * npk):
        assert check_rank(r, npi, npk, g, bi[r], bk[r], ri[r], rk[r], a[r]), r


def _worker(rank, world, port, npi, npk, nx, ny, nz, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cw, cx, cz = dist_comms(npi, npk)
        g = torch.arange(nx * ny * nz, dtype=torch.float64).view(nz, ny, nx) + 0.5
        a = {rank: blocks(g, npi, npk)[rank]}
        imax, kmax = nx // npi, nz // npk
        bi = trp_i_forward(cx, a, imax, ny * kmax)
        bk = trp_k_forward(cz, a, imax * ny, kmax)
        ri = trp_i_backward(cx, bi, imax, ny * kmax)
        rk = trp_k_backward(cz, bk, imax * ny, kmax)
        ok = check_rank(rank, npi, npk, g, bi[rank], bk[rank], ri[rank], rk[rank], a[rank])
        # the uneven all-to-all of the kx pencils over the world communicator, keyed by world rank
        P = world
        cnt = [p + 1 for p in range(P)]
        send = {rank: torch.cat([torch.full((c,), 10.0 * rank + p) for p, c in enumerate(cnt)])}
        recv = {rank: torch.zeros((rank + 1) * P)}
        cw.all_to_all_v(send, {rank: cnt}, recv, {rank: [rank + 1] * P}).wait()
        ok = ok and torch.equal(recv[rank], torch.cat([torch.full((rank + 1,), 10.0 * p + rank) for p in range(P)]))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_cartesian_transposes_four_processes_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 2, 2, 8, 3, 8, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1, 2, 3]
    assert all(r[1] for r in res), res
