#!/usr/bin/env python3
"""Times OPR_Poisson (with dpdy) on one GPU."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tlab_amd as T

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=512); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--direct", action="store_true", help="EllipticOrder=CompactDirect6 (y tables of tests/golden/direct_y.npz: n = 24, 64, 128, 512)")
args = ap.parse_args(); n = args.n
T.init(0)
x = np.arange(n) / n; y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(2))
gx, gy, gz = T.FdmPlan(x, True, True), T.FdmPlan(y, False, False), T.FdmPlan(x, True, True)
torch.cuda.synchronize(); import time; t0 = time.time()
gy_ell = None
if args.direct:
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "direct_y.npz"))
    tab = {k[len("ny%d_" % n):]: g[k] for k in g.files if k.startswith("ny%d_" % n)}
    gy = gy_ell = T.FdmPlan.from_tables(tab, scheme1=6, scheme2=16)
plan = T.PoissonPlan(gx, gy, gz, n, n, n, gy_elliptic=gy_ell)
torch.cuda.synchronize(); print("plan creation %.2f s" % (time.time() - t0))
N = n ** 3
f = torch.rand(N, dtype=torch.float64, device="cuda") - 0.5
hb = torch.zeros(n * n, dtype=torch.float64, device="cuda"); ht = torch.zeros_like(hb)
t1 = torch.empty(plan.isize_txc_field, dtype=torch.float64, device="cuda"); t2 = torch.empty_like(t1)
p = torch.empty_like(f); d = torch.empty_like(f)
ts = []
for it in range(args.iters + 1):
    p.copy_(f); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); T.OPR_Poisson(plan, n, n, n, T.BCS_NN, p, t1, t2, hb, ht, d); b.record(); torch.cuda.synchronize()
    if it: ts.append(a.elapsed_time(b))
ts.sort(); med = ts[len(ts) // 2]
print("OPR_Poisson %d^3: %.3f ms  %.3e pts/s  %.1f GB/s alg (24 B/pt)  mem %.1f GB" % (n, med, N / med * 1e3, 24 * N / med / 1e6, torch.cuda.memory_allocated() / 1e9))
if os.environ.get("TLAB_PROFILE_REPORT"):
    import ctypes
    from tlab_amd.lib import load
    L = load(); L.tlab_profile_reset(); L.tlab_profile_enable(1)
    for it in range(5):
        p.copy_(f); T.OPR_Poisson(plan, n, n, n, T.BCS_NN, p, t1, t2, hb, ht, d)
    torch.cuda.synchronize(); L.tlab_profile_enable(0)
    buf = ctypes.create_string_buffer(8192); L.tlab_profile_report(buf, 8192); print(buf.value.decode())
