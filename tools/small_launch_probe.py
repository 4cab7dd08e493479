#!/usr/bin/env python3
"""How much of the decomposition overhead of 8 z-slabs is the SIZE of the launches?  OPR_Burgers_Y / _X (one field) and a plain copy on 512 x 512 x nz boxes,
nz = 512 (the single domain) and nz = 64 (one of eight slabs): time per launch and per point."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tlab_amd as T
T.init(0)
nx = ny = 512
x = np.arange(nx) / nx
gx, gy = T.FdmPlan(x, True, True), T.FdmPlan(np.arange(ny) / (ny - 1.0), False, True)


def med(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


for nz in (512, 256, 128, 64, 32):
    N = nx * ny * nz
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    u = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
    v = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5
    r, t = torch.empty_like(u), torch.empty_like(u)
    tc = med(lambda: r.copy_(u))
    ty = med(lambda: T.OPR_Burgers_Y(T.OPR_B_U_IN, 1e-3, nx, ny, nz, 0, gy, u, v, r, t))
    tx = med(lambda: T.OPR_Burgers_X(T.OPR_B_U_IN, 1e-3, nx, ny, nz, 0, gx, u, v, r, t))
    tp = med(lambda: T.OPR_Partial_Y(T.OPR_P1, nx, ny, nz, 0, gy, u, r, None))
    print("nz %4d: copy %7.1f us = %5.0f GB/s | Burgers_Y %7.1f us = %5.0f GB/s (24 B/pt) | Burgers_X %7.1f us = %5.0f GB/s | Partial_Y(P1) %7.1f us = %5.0f GB/s (16 B/pt)"
          % (nz, tc * 1e3, 16 * N / tc / 1e6, ty * 1e3, 24 * N / ty / 1e6, tx * 1e3, 24 * N / tx / 1e6, tp * 1e3, 16 * N / tp / 1e6))

# the same launches COLD: eight sets of arrays visited in turn (4.3 GB at nz = 64: nothing of a set survives in the 256-MiB Infinity Cache until its next turn)
print("cold (8 sets of arrays in turn):")
for nz in (64, 128):
    N = nx * ny * nz
    sets = []
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    for k in range(8):
        sets.append([torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5 for _ in range(2)] + [torch.empty(N, dtype=torch.float64, device="cuda") for _ in range(2)])
    state = {"k": 0}

    def nxt():
        state["k"] = (state["k"] + 1) % 8
        return sets[state["k"]]

    def fy():
        u, v, r, t = nxt(); T.OPR_Burgers_Y(T.OPR_B_U_IN, 1e-3, nx, ny, nz, 0, gy, u, v, r, t)

    def fx():
        u, v, r, t = nxt(); T.OPR_Burgers_X(T.OPR_B_U_IN, 1e-3, nx, ny, nz, 0, gx, u, v, r, t)

    def fc():
        u, v, r, t = nxt(); r.copy_(u)
    tc, ty, tx = med(fc, 48), med(fy, 48), med(fx, 48)
    print("nz %4d: copy %7.1f us = %5.0f GB/s | Burgers_Y %7.1f us = %5.0f GB/s | Burgers_X %7.1f us = %5.0f GB/s" % (nz, tc * 1e3, 16 * N / tc / 1e6, ty * 1e3, 24 * N / ty / 1e6, tx * 1e3, 24 * N / tx / 1e6))
    del sets

# the three-field accumulate launch of the RHS drivers (tlab_opr_burgers_add_n, y direction), cold, per slab size
import ctypes
from tlab_amd.lib import load, c_vp, check
L = load()
print("three fields + accumulate (k_htile<BURGERS>, 80 B per point), cold:")
for nz in (512, 64):
    N = nx * ny * nz
    nsets = 2 if nz == 512 else 8
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    sets = [[torch.rand(N, dtype=torch.float64, device="cuda", generator=g) - 0.5 for _ in range(3)] + [torch.zeros(N, dtype=torch.float64, device="cuda") for _ in range(5)] for _ in range(nsets)]
    state = {"k": 0}
    nu = (ctypes.c_double * 3)(1e-3, 1e-3, 1e-3)

    def f3():
        state["k"] = (state["k"] + 1) % nsets
        a = sets[state["k"]]
        s = (c_vp * 3)(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr())
        r = (c_vp * 3)(a[3].data_ptr(), a[4].data_ptr(), a[5].data_ptr())
        check(L.tlab_opr_burgers_add_n(2, gy._h, nx, ny, nz, 0, 3, nu, s, a[1].data_ptr(), r, a[6].data_ptr(), a[7].data_ptr(), 0), "add_n")
    t3 = med(f3, 32)
    print("nz %4d: %7.1f us = %5.0f GB/s" % (nz, t3 * 1e3, 80 * N / t3 / 1e6))
    del sets
