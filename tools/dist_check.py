#!/usr/bin/env python3
"""Multi-process check of the z-slab algorithm: every rank runs its slab through SlabDns(DistComm) and compares with the
single-domain substep computed redundantly on its own device.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port 29533 tools/dist_check.py [--zmode halo]

On an N-GPU node it uses RCCL (backend nccl).  With TLAB_DIST_BACKEND=gloo all ranks may share one GPU (payloads staged through
the host): that is how the multi-process logic is exercised on the single-GPU test box (tests/test_gpu_dist.py)."""
import argparse
import os
import sys

import numpy as np

REF_HYPER = 0.1      # wall closure of the flang-built reference (defect 1); the driver classes default to the consistent 0.0

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=32)
    ap.add_argument("--ny", type=int, default=24)
    ap.add_argument("--nz", type=int, default=128)
    ap.add_argument("--zmode", default="auto")
    ap.add_argument("--bcs", default="noslip")
    ap.add_argument("--driver", default="python", choices=["python", "native"], help="python: tlab_amd/parallel.py::SlabDns over torch.distributed; "
                    "native: the C++ driver (tlab_slab_dns_*) with the RCCL transport of libtlab_amd_comm.so (backend nccl) or, on gloo, with the "
                    "host-staged callback transport (several ranks on one GPU)")
    ap.add_argument("--cases", default="", help='several cases in one launch: "driver:zmode:nz:bcs:nx;..." (overrides the single-case options)')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TLAB_DIST_BACKEND", "nccl")
    dev = local % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    bootstrap = os.environ.get("TLAB_DIST_BOOTSTRAP", backend)      # gloo with backend nccl: torch.distributed carries the ncclUniqueId and the control
    if bootstrap == "nccl":                                          # reductions only, the RCCL of libtlab_amd_comm.so is the only RCCL user (bench.py's default)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(bootstrap)
    import tlab_amd as T
    from tlab_amd.dns import Dns
    from tlab_amd.parallel import SlabDns, DistComm
    T.init(dev)
    # --cases "driver:zmode:nz:bcs:nx;..." : several checks in ONE launch (one interpreter start-up and one process group per world size instead of one
    # per case: tests/test_gpu_dist.py); without it the single case of the other options
    cases = [(args.driver, args.zmode, args.nz, args.bcs, args.nx)]
    if args.cases:
        cases = []
        for c in args.cases.split(";"):
            drv, zm, nz_, bcs_, nx_ = c.split(":")
            cases.append((drv, zm, int(nz_), bcs_, int(nx_)))
    overall = 0.0
    for drv, zm, nz, bcs, nx in cases:
        worst = one_case(args, T, Dns, SlabDns, DistComm, torch, dist, rank, world, backend, bootstrap, drv, zm, nx, args.ny, nz, bcs)
        overall = max(overall, worst)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if overall <= 1e-11 else 1)


def one_case(args, T, Dns, SlabDns, DistComm, torch, dist, rank, world, backend, bootstrap, driver, zmode, nx, ny, nz, bcs):
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    rng = np.random.default_rng(7)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    fields = [((np.sin(np.pi * X + k) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(4)]
    one = Dns(x, y, z, nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    if driver == "native":
        from tlab_amd.slab import NativeSlabDns
        slab = NativeSlabDns("rccl" if backend == "nccl" else "dist", x, y, z, nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, hyper_bc1_ext=REF_HYPER)
    else:
        slab = SlabDns(DistComm(), x, y, z, nscal=1, visc=1.0 / 600.0, schmidt=(0.8,), yuniform=False, zmode=zmode, hyper_bc1_ext=REF_HYPER)
    if bcs == "freeslip":
        one.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
        slab.set_bcs("freeslip", "freeslip", "neumann", "dirichlet")
    for i in range(3):
        t = torch.from_numpy(fields[i]).cuda()
        one.q[i].copy_(t); slab.scatter("q", i, t)
    t = torch.from_numpy(fields[3]).cuda()
    one.s[0].copy_(t); slab.scatter("s", 0, t)
    dtime = 2e-3
    for k in range(3):
        if k == 0:
            for h in one.hq + one.hs:
                h.zero_()
        last = k == 2
        one.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * one.kdt[k], 1.0 if last else one.kco[k], not last)
        slab.substep_of_cycle(k, dtime)
    worst = 0.0
    n = slab.n
    for name, ref in (("q", one.q), ("hq", one.hq), ("s", one.s), ("hs", one.hs)):
        for i, rf in enumerate(ref):
            got = slab.st[rank][name][i]
            err = float((got - rf[rank * n:(rank + 1) * n]).abs().max() / rf.abs().max())
            worst = max(worst, err)
    # per-iteration monitors on slabs: the same maxima as the single domain (MPI_MAX / MPI_MIN of two scalars, time.f90:522)
    (a1, a2), dta = one.TIME_COURANT(1.2, 0.3)
    (b1, b2), dtb = slab.TIME_COURANT(1.2, 0.3)
    worst = max(worst, abs(a1 - b1) / a1, abs(a2 - b2) / a2, abs(dta - dtb) / dta)
    dmin, dmax = one.dilatation_bounds()
    smin, smax = slab.dilatation_bounds()
    worst = max(worst, abs(dmin - smin) / a1, abs(dmax - smax) / a1)      # a1 = max(|u_i|/h_i): the size of the terms of div(q)
    tt = torch.tensor([worst], dtype=torch.float64)
    if bootstrap == "nccl":
        tt = tt.cuda()
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    rccl_ranks = None
    nc = getattr(slab, "_keep", None)
    if driver == "native" and backend == "nccl" and nc is not None and hasattr(nc, "info"):
        rccl_ranks = nc.info(6)            # ncclCommCount of the communicator inside libtlab_amd_comm.so
    if rank == 0:
        print("DIST_CHECK driver=%s world=%d zmode=%s backend=%s bootstrap=%s fused_x=%d nx=%d nz=%d bcs=%s rccl_ranks=%s worst_rel_err=%.3e %s" % (
            driver, world, slab.zmode, backend, bootstrap, int(getattr(slab, "fused_x", False)), nx, nz, bcs, rccl_ranks, float(tt.item()),
            "OK" if float(tt.item()) <= 1e-11 else "FAIL"), flush=True)
    torch.cuda.synchronize()
    if hasattr(slab, "close"):
        slab.close()
    dist.barrier()
    return float(tt.item())


if __name__ == "__main__":
    main()
