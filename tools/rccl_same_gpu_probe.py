#!/usr/bin/env python3
"""Probe: can two torch.distributed ranks on backend nccl (= RCCL) share ONE GPU on this box?  (If so, the RCCL legs of the slab driver can
be exercised with real inter-process collectives on a single-GPU box instead of the host-staged gloo stand-in.)"""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        x = torch.arange(4, dtype=torch.float64, device="cuda") + 10.0 * rank
        y = torch.empty_like(x)
        dist.all_to_all_single(y, x)
        torch.cuda.synchronize()
        print("rank", rank, "all_to_all_single ok:", y.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29617), nprocs=2, join=True)
