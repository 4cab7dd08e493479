#!/usr/bin/env python3
"""The four DistComm operations of the slab driver over RCCL (backend nccl), device buffers, no host staging.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port 29551 tools/rccl_check.py

Runs with any P <= number of GPUs.  On the single-GPU test box P = 1: every peer is the rank itself, which still goes through
ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd and the stream ordering between the collectives and the library's kernels."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import torch.distributed as dist
    rank, P, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from tlab_amd.parallel import DistComm
    comm = DistComm()
    assert not comm.stage_host
    dev = torch.device("cuda", local)
    bad = []
    tag = lambda src, dst, i: 1000.0 * src + 10.0 * dst + i          # noqa: E731

    # all_to_all: row p of the send buffer goes to rank p
    m = 4096
    s = torch.stack([torch.full((m,), tag(rank, p, 0), dtype=torch.float64, device=dev) for p in range(P)])
    got = comm.all_to_all({rank: s})[rank]
    want = torch.stack([torch.full((m,), tag(p, rank, 0), dtype=torch.float64, device=dev) for p in range(P)])
    if not torch.equal(got, want):
        bad.append("all_to_all")

    # all_to_all_v with uneven counts (the kx-pencils: rank p owns 3 + p columns)
    cnt = lambda src, dst: 64 * (3 + dst) * (2 + src)               # noqa: E731
    sc, rc = [cnt(rank, p) for p in range(P)], [cnt(p, rank) for p in range(P)]
    send = torch.cat([torch.full((sc[p],), tag(rank, p, 1), dtype=torch.float64, device=dev) for p in range(P)])
    recv = torch.zeros(sum(rc), dtype=torch.float64, device=dev)
    comm.all_to_all_v({rank: send}, {rank: sc}, {rank: recv}, {rank: rc}).wait()
    want = torch.cat([torch.full((rc[p],), tag(p, rank, 1), dtype=torch.float64, device=dev) for p in range(P)])
    if not torch.equal(recv, want):
        bad.append("all_to_all_v")

    # neighbour ring, two messages each way, consumed by a kernel on the current stream right after wait()
    left, right = (rank - 1) % P, (rank + 1) % P
    tl = [torch.full((m,), tag(rank, left, 2 + i), dtype=torch.float64, device=dev) for i in range(2)]
    tr = [torch.full((m,), tag(rank, right, 4 + i), dtype=torch.float64, device=dev) for i in range(2)]
    fr = [torch.zeros(m, dtype=torch.float64, device=dev) for _ in range(2)]
    fl = [torch.zeros(m, dtype=torch.float64, device=dev) for _ in range(2)]
    comm.neighbor_exchange({rank: tl}, {rank: tr}, {rank: fr}, {rank: fl}).wait()
    for i in range(2):
        if not bool((fr[i] == tag(right, rank, 2 + i)).all()):       # what my right neighbour sent to ITS left
            bad.append("neighbor_exchange from right %d" % i)
        if not bool((fl[i] == tag(left, rank, 4 + i)).all()):
            bad.append("neighbor_exchange from left %d" % i)

    mx = comm.all_reduce({rank: [float(rank), -float(rank)]}, "max")
    mn = comm.all_reduce({rank: [float(rank), -float(rank)]}, "min")
    if mx != [float(P - 1), 0.0] or mn != [0.0, -float(P - 1)]:
        bad.append("all_reduce")

    t = torch.tensor([float(len(bad))], device=dev)
    dist.all_reduce(t)
    if rank == 0:
        print("RCCL_CHECK world=%d %s %s" % (P, "OK" if t.item() == 0 else "FAIL", bad))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if t.item() == 0 else 1)


if __name__ == "__main__":
    main()
