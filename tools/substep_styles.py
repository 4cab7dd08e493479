import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import tlab_amd as T
from tlab_amd.dns import Dns
T.init(0)
n = 512
x = np.arange(n) / n; y = np.arange(n) / (n - 1.0)
d = Dns(x, y, x.copy(), nscal=1, visc=1/5000., schmidt=(1.0,), yuniform=True, hyper_bc1_ext=0.0)
for t in d.q + d.s: t.copy_(0.1 * (torch.rand_like(t) - 0.5))
dt = 1e-3
def fused(k):
    d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dt * d.kdt[k % 3], d.kco[k % 3] if k % 3 < 2 else 1.0, k % 3 < 2)
def host_style(k):      # what the unchanged time.f90 does on device arrays: RHS, DAXPY x 4, DSCAL x 4
    dte = dt * d.kdt[k % 3]
    d.RHS_GLOBAL_INCOMPRESSIBLE_1(dte)
    for qq, hh in zip(d.q + d.s, d.hq + d.hs): qq.add_(hh, alpha=dte)
    if k % 3 < 2:
        for hh in d.hq + d.hs: hh.mul_(d.kco[k % 3])
for name, f in (("fused substep", fused), ("RHS + DAXPY + DSCAL", host_style), ("fused substep", fused), ("RHS + DAXPY + DSCAL", host_style)):
    for k in range(3): f(k)
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(12): f(k)
    torch.cuda.synchronize(); print(name, "%.2f ms per substep" % ((time.time() - t0) / 12 * 1e3))
