#!/usr/bin/env python3
"""x-line kernels (k_xline) at the line lengths of BASELINE.json's configs: OPR_Partial_X (P1, P2_P1), OPR_Burgers_X, and the 4-field fused
Burgers launch of the RHS driver.  Prints ms, grid points/s and algorithmic GB/s (16 / 24 / 24 B per point; 4 fields: 8 + 3*16 + 8 B = 13 arrays
of 8 B: velocity once, 3 other operands, 4 old + 4 new tendencies).  TLAB_XLINE_WIDE=0 selects the one-wave-per-line forms for lines of 1024 /
2048 points (A/B of the several-waves-per-line forms)."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tlab_amd as T  # noqa: E402
from tlab_amd.lib import load, check, c_vp  # noqa: E402


def timeit(fn, iters=12, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="512x512x512,1024x512x128,2048x1024x64")
    ap.add_argument("--exact-uniform", action="store_true", help="A/B: a plan whose table rows are all the middle row of the generated ones (every "
                    "chunk then has the same tables: scalar loads instead of per-lane tables in LDS)")
    args = ap.parse_args()
    T.init(0)
    L = load()
    out = []
    for spec in args.grids.split(","):
        nx, ny, nz = (int(v) for v in spec.split("x"))
        N = nx * ny * nz
        g = T.FdmPlan(np.arange(nx) / nx * 2.0, True, True)
        if args.exact_uniform:
            tabs = [np.tile(g.table(k)[nx // 2], (nx, 1)) for k in ("lhs1", "rhs1", "lhs2", "rhs2")]
            g = T.FdmPlan.from_arrays(nx, True, 0, tabs[0], tabs[1][:, :g.info(2)], tabs[2], tabs[3][:, :g.info(4) + 3])
        gen = torch.Generator(device="cuda"); gen.manual_seed(nx)
        f = [torch.rand(N, dtype=torch.float64, device="cuda", generator=gen) for _ in range(4)]
        h = [torch.zeros(N, dtype=torch.float64, device="cuda") for _ in range(4)]
        tmp = torch.empty(N, dtype=torch.float64, device="cuda")
        tmp2 = torch.empty(N, dtype=torch.float64, device="cuda")
        rec = {"grid": [nx, ny, nz], "wide": os.environ.get("TLAB_XLINE_WIDE", "1"), "chunks": g.info(8), "same_tables_in_every_chunk": g.info(9)}
        ms = timeit(lambda: T.OPR_Partial_X(T.OPR_P1, nx, ny, nz, 0, g, f[0], h[0], tmp))
        rec["P1"] = {"ms": ms, "GBps": 16.0 * N / ms / 1e6}
        ms = timeit(lambda: T.OPR_Partial_X(T.OPR_P2_P1, nx, ny, nz, 0, g, f[0], h[0], tmp))
        rec["P2_P1"] = {"ms": ms, "GBps": 24.0 * N / ms / 1e6}
        ms = timeit(lambda: T.OPR_Burgers_X(T.OPR_B_U_IN, 2e-4, nx, ny, nz, 0, g, f[1], f[0], h[0], tmp))
        rec["Burgers"] = {"ms": ms, "GBps": 24.0 * N / ms / 1e6}
        nu = (ctypes.c_double * 4)(2e-4, 2e-4, 2e-4, 2e-4)
        sp = (c_vp * 4)(*[t.data_ptr() for t in f])
        hp = (c_vp * 4)(*[t.data_ptr() for t in h])

        def four():
            check(L.tlab_opr_burgers_add_n(1, g._h, nx, ny, nz, 0, 4, nu, sp, f[0].data_ptr(), hp, tmp.data_ptr(), tmp2.data_ptr(), 0), "burgers_add_n")
        ms = timeit(four)
        rec["Burgers_x4_acc"] = {"ms": ms, "GBps": (8.0 + 3 * 8.0 + 4 * 16.0) * N / ms / 1e6}
        rec["path"] = L.tlab_last_kernel_path()
        out.append(rec)
        print(json.dumps(rec), flush=True)
        del f, h, tmp, tmp2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
