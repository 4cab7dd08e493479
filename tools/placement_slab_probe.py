#!/usr/bin/env python3
"""Does the set of allocations matter for the z-slab driver as it does for the single domain (tools/placement_substep.py)?  All P slab ranks of the n^3
box on one GPU (loopback transport); the substep with the arrays as allocated, then with every rank's 17 arrays re-drawn at random from a pool of its
own, several times.       python tools/placement_slab_probe.py [n] [P] [trials]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import tlab_amd as T
    from tlab_amd.slab import NativeSlabDns
    from tlab_amd.lib import load, c_vp, check
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    T.init(0)
    L = load()
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    d = NativeSlabDns("loopback", x, y, x, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, hyper_bc1_ext=0.0, size=P)
    rng = np.random.default_rng(3)
    m = d.isize_txc
    pools = {r: [torch.zeros(m, dtype=torch.float64, device="cuda") for _ in range(34)] for r in d.local_ranks}

    def bind(l, arrs):
        arr = lambda ts: (c_vp * max(len(ts), 1))(*[t.data_ptr() for t in ts])       # noqa: E731
        check(L.tlab_slab_dns_bind(d._h, l, arr(arrs[0:3]), arr(arrs[3:4]), arr(arrs[4:7]), arr(arrs[7:8]), arr(arrs[8:17])), "tlab_slab_dns_bind")

    def run(label):
        for k in range(3):
            d.substep_of_cycle(k, 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(9):
            d.substep_of_cycle(k, 1e-3)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 9 * 1e3
        print("%-44s %.3f ms per substep" % (label, ms), flush=True)
        return ms

    run("arrays as NativeSlabDns allocated them")
    run("again")
    res = []
    for t in range(trials):
        for l, r in enumerate(d.local_ranks):
            idx = rng.permutation(34)[:17]
            bind(l, [pools[r][i] for i in idx])
        res.append(run("random assignment %d (every rank re-drawn)" % t))
    print("min %.3f  median %.3f  max %.3f" % (min(res), sorted(res)[len(res) // 2], max(res)))


if __name__ == "__main__":
    main()
