#!/usr/bin/env python3
"""Timings of the REFERENCE'S OWN Fortran (oracle/_ref: flang -O2, compiled where it lies) per operator, in the build container (no GPU): the numbers
BASELINE.md section 2 carries (its section 3, item 1).  Serial build on 1 core; with TLAB_REF_LIB pointing at an OpenMP build (make -C oracle omp)
and OMP_NUM_THREADS set, the reference's OpenMP loops on the container's cores.

    python tools/time_ref.py [128 256]          # Mpts/s per call, median of 5 after 1 warm-up
TEST INFRASTRUCTURE ONLY (uses oracle/)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from oracle import ref_lib as R
    if os.environ.get("TLAB_REF_LIB"):
        R._PATH = os.environ["TLAB_REF_LIB"]
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing (make -C oracle)")
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256]
    threads = os.environ.get("OMP_NUM_THREADS", "1")
    print("reference Fortran (%s), OMP_NUM_THREADS=%s, %d CPUs visible" % (R._PATH, threads, os.cpu_count()))
    for n in sizes:
        x = np.arange(n) / n
        y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2))
        R.init(n, n, n)
        R.fdm_create(1, x, True, True)
        R.fdm_create(2, y, False, False)
        R.fdm_create(3, x, True, True)
        rng = np.random.default_rng(n)
        u = rng.uniform(-1, 1, n ** 3)
        v = rng.uniform(-1, 1, n ** 3)
        r, t = np.zeros_like(u), np.zeros_like(u)
        L = R.lib()
        rows = []
        for d in (1, 2, 3):
            for name, typ in (("OPR_P1", 1), ("OPR_P2_P1", 3)):
                ts = []
                for it in range(6):
                    t0 = time.perf_counter()
                    L.ref_partial(d, typ, n, n, n, 0, u, r, t)
                    ts.append(time.perf_counter() - t0)
                rows.append(("OPR_Partial_%s(%s)" % ("XYZ"[d - 1], name), sorted(ts[1:])[2]))
            ts = []
            for it in range(4):
                t0 = time.perf_counter()
                L.ref_burgers(d, n, n, n, 0, 1e-3, u, v, r, t)
                ts.append(time.perf_counter() - t0)
            rows.append(("OPR_Burgers_%s (transcription on FDM_Der1/2_Solve)" % "XYZ"[d - 1], sorted(ts[1:])[1]))
        for name, sec in rows:
            print("%4d^3  %-58s %8.3f s  %8.1f Mpts/s" % (n, name, sec, n ** 3 / sec / 1e6))


if __name__ == "__main__":
    main()
