#!/usr/bin/env python3
"""Host time to ENQUEUE one substep of the native z-slab driver (tlab_slab_dns_substep), per rank: P loopback ranks in this process, on torch's
default stream and on a side stream; the GPU finishes later (the difference to the wall time per step is device time).
    python tools/slab_host_issue.py [P] [n] [nz]        (P = 1, nz = 64: one rank's share of the 8-GPU run, its own ring neighbour)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import tlab_amd as T
    from tlab_amd.slab import NativeSlabDns
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nz = int(sys.argv[3]) if len(sys.argv) > 3 else n
    T.init(0)
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    d = NativeSlabDns("loopback", x, y, np.arange(nz) / nz, size=P, nscal=1)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for r in range(P):
        for t in d.st[r]["q"] + d.st[r]["s"]:
            t.copy_(0.1 * (torch.rand(t.numel(), dtype=torch.float64, device="cuda", generator=gen) - 0.5))
    side = torch.cuda.Stream()
    for name, ctx in (("default stream", torch.cuda.stream(torch.cuda.default_stream())), ("side stream", torch.cuda.stream(side))):
        with ctx:
            for k in range(3):
                d.substep_of_cycle(k, 1e-3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(12):
                d.substep_of_cycle(k, 1e-3)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("%s: host issue %.3f ms per substep of all %d ranks = %.3f ms per rank; wall %.2f ms per substep" %
              (name, (t1 - t0) / 12 * 1e3, P, (t1 - t0) / 12 / P * 1e3, (t2 - t0) / 12 * 1e3))


if __name__ == "__main__":
    main()
