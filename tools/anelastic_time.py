#!/usr/bin/env python3
"""Times the substep with the anelastic density weights (literal operator sequence) beside the incompressible one at 512^3."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tlab_amd as T
from tlab_amd.dns import Dns
T.init(0)
n = 512
x = np.arange(n) / n; y = np.arange(n) / (n - 1.0)
for anel in (False, True, False, True):
    d = Dns(x, y, x.copy(), nscal=1, visc=1/5000., schmidt=(1.0,), yuniform=True, hyper_bc1_ext=0.0)
    if anel:
        rb = 1.0 + 0.3 * np.exp(-2.0 * y)
        d.set_anelastic(rb, 1.0 / rb)
    for t in d.q + d.s: t.copy_(0.1 * (torch.rand_like(t) - 0.5))
    dt = 1e-3
    def sub(k): d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dt * d.kdt[k % 3], d.kco[k % 3] if k % 3 < 2 else 1.0, k % 3 < 2)
    for k in range(3): sub(k)
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(9): sub(k)
    torch.cuda.synchronize(); print("anelastic" if anel else "incompressible", "%.2f ms per substep" % ((time.time() - t0) / 9 * 1e3))
    del d
if os.environ.get("TLAB_PROFILE_REPORT"):
    import ctypes
    from tlab_amd.lib import load
    d = Dns(x, y, x.copy(), nscal=1, visc=1/5000., schmidt=(1.0,), yuniform=True, hyper_bc1_ext=0.0)
    rb = 1.0 + 0.3 * np.exp(-2.0 * y); d.set_anelastic(rb, 1.0 / rb)
    for t in d.q + d.s: t.copy_(0.1 * (torch.rand_like(t) - 0.5))
    L = load(); L.tlab_profile_reset(); L.tlab_profile_enable(1)
    for k in range(3): d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(1e-3 * d.kdt[k], d.kco[k] if k < 2 else 1.0, k < 2)
    torch.cuda.synchronize(); L.tlab_profile_enable(0)
    buf = ctypes.create_string_buffer(16384); L.tlab_profile_report(buf, 16384)
    for r in buf.value.decode().splitlines():
        c = r.split("\t")
        if len(c) >= 3: print("%-32s %4s calls %8.3f ms per substep" % (c[0], c[1], float(c[2]) / 3))
