#!/usr/bin/env python3
"""How much of the RK substep's time depends on WHICH allocations play the roles of q, s, hq, hs, txc?  (tools/placement_probe: a 13-stream kernel runs at
4.8 .. 5.9 TB/s depending on the set of 1-GiB allocations it streams, reproducibly per set -- none of the arrays is slow by itself.)
A pool of P allocations of the txc size; trial 0 takes them in allocation order, the others a random assignment; per trial: ms per substep and the
average launch time of the bandwidth-bound kernels.      python tools/placement_substep.py [n] [pool] [trials]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import tlab_amd as T
    from tlab_amd.dns import Dns, RKM_EXP3
    from tlab_amd.lib import load
    import bench
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    T.init(0)
    L = load()
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    d = Dns(x, y, x, nscal=1, visc=1.0 / 5000.0, schmidt=(1.0,), yuniform=True, rkm_mode=RKM_EXP3, hyper_bc1_ext=0.0)
    m = d.isize_txc_field
    ref = [t.clone() for t in d.q + d.s]
    bench.synthetic_fields(ref, n, n, n, 0, n, 0)
    own = d.q + d.s + d.hq + d.hs + d.txc
    pool = [torch.zeros(m, dtype=torch.float64, device="cuda") for _ in range(P)]
    rng = np.random.default_rng(7)
    names = ["k_xline<BURGERS>", "k_htile<BURGERS>", "k_ptile<BURGERS>", "k_htile<BURGERS+div>", "k_ptile<BURGERS+div>", "k_xline<P1>", "k_htile<P1>", "k_fftx_c2r<final>"]
    buf = ctypes.create_string_buffer(1 << 16)

    def run(arrs, label):
        d.q, d.s, d.hq, d.hs = [a[: d.n] for a in arrs[0:3]], [arrs[3][: d.n]], [a[: d.n] for a in arrs[4:7]], [arrs[7][: d.n]]
        d.txc = [a[:m] if a.numel() >= m else a for a in arrs[8:17]]
        d._ptrs = None
        for t, r in zip(d.q + d.s, ref):
            t.copy_(r)
        d.TIME_RUNGEKUTTA(1e-3)
        L.tlab_profile_reset()
        L.tlab_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            d.TIME_RUNGEKUTTA(1e-3)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 9 * 1e3
        L.tlab_profile_enable(0)
        L.tlab_profile_report(buf, len(buf))
        ks = {}
        for line in buf.value.decode().strip().split("\n"):
            p = line.split("\t")
            if len(p) >= 3:
                ks[p[0]] = float(p[2]) / max(int(p[1]), 1)
        print("%-34s ms_per_substep %.3f  " % (label, ms) + "  ".join("%s %.3f" % (k.replace("BURGERS", "B"), ks.get(k, float("nan"))) for k in names), flush=True)
        if os.environ.get("PLACEMENT_LOG"):
            with open(os.environ["PLACEMENT_LOG"], "a") as f:
                f.write(json.dumps({"label": label, "ms": ms, "ptr": [int(a.data_ptr()) for a in arrs[:17]], "kernels": ks}) + "\n")
        return ms

    print(buf.value.decode()[:0], end="")
    run(own, "the driver's own 17 allocations")
    run(own, "the same again")
    res = []
    for t in range(trials):
        idx = list(range(17)) if t == 0 else [int(i) for i in rng.permutation(P)[:17]]
        ms = run([pool[i] for i in idx], "pool in allocation order" if t == 0 else "random assignment %d %s" % (t, ",".join(str(i) for i in idx)))
        res.append((ms, idx))
    res.sort()
    print("best %.3f  median %.3f  worst %.3f ms per substep over %d assignments" % (res[0][0], res[len(res) // 2][0], res[-1][0], len(res)))
    best = res[0][1]
    run([pool[i] for i in best], "best assignment again")
    run(own, "the driver's own allocations again")


if __name__ == "__main__":
    main()
