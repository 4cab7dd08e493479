#!/usr/bin/env python3
"""Per-operator micro-benchmark on one GPU: time, grid points/s and algorithmic-bytes bandwidth
(16 B/pt for P1/P2, 24 B/pt for P2_P1 and Burgers)."""
import argparse
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tlab_amd as T  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rtile-m", type=int, default=0)
    ap.add_argument("--dirs", default="123")
    ap.add_argument("--types", default="P1,P2,P2_P1,BURGERS", help="which operators to time (P1 alone: the north-star's own kernels, one template each under rocprofv3)")
    args = ap.parse_args()
    n = args.n
    T.init(0)
    if args.rtile_m:
        T.load().tlab_set_tuning(1, args.rtile_m)
    N = n ** 3
    x = np.arange(n) / n
    y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2))
    plans = {1: T.FdmPlan(x, True, True), 2: T.FdmPlan(y, False, False), 3: T.FdmPlan(x, True, True),
             "2u": T.FdmPlan(np.arange(n) / (n - 1.0), False, True)}
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    u = torch.rand(N, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    v = torch.rand(N, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    r, t = torch.empty_like(u), torch.empty_like(u)
    # reference point: plain device copy (read + write = 16 B/pt)
    med, best = timeit(lambda: r.copy_(u), args.iters)
    print("%-28s %8.3f ms  %7.1f GB/s (16 B/pt)" % ("torch copy", med, 16 * N / med / 1e6))
    part = {1: T.OPR_Partial_X, 2: T.OPR_Partial_Y, 3: T.OPR_Partial_Z}
    burg = {1: T.OPR_Burgers_X, 2: T.OPR_Burgers_Y, 3: T.OPR_Burgers_Z}
    rows = []
    for d in (1, 2, 3):
        if str(d) not in args.dirs:
            continue
        keys = [d] + (["2u"] if d == 2 else [])
        for key in keys:
            g = plans[key]
            tag = "XYZ"[d - 1] + ("(uniform)" if key == "2u" else "")
            for name, typ, bpp in (("P1", T.OPR_P1, 16), ("P2", T.OPR_P2, 16), ("P2_P1", T.OPR_P2_P1, 24)):
                if name not in args.types.split(","):
                    continue
                med, best = timeit(lambda: part[d](typ, n, n, n, 0, g, u, r, t), args.iters)
                rows.append(("OPR_Partial_%s %s" % (tag, name), med, best, bpp))
            if "BURGERS" in args.types.split(","):
                med, best = timeit(lambda: burg[d](T.OPR_B_U_IN, 1e-3, n, n, n, 0, g, u, v, r, t), args.iters)
                rows.append(("OPR_Burgers_%s U_IN" % tag, med, best, 24))
    for name, med, best, bpp in rows:
        print("%-28s %8.3f ms (best %7.3f)  %9.3e pts/s  %7.1f GB/s alg (%d B/pt)  %5.1f %% of 8 TB/s" %
              (name, med, best, N / med * 1e3, bpp * N / med / 1e6, bpp, bpp * N / med / 1e6 / 80.0))


if __name__ == "__main__":
    main()
