"""Diagnostic: host-side enqueue time of the z-slab driver per substep (all ranks simulated in this process) against the device time."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tlab_amd as T
from tlab_amd.parallel import SlabDns, LoopbackComm
T.init(0)
n = 512; P = 8
x = np.arange(n)/n; y = np.arange(n)/(n-1.0)
d = SlabDns(LoopbackComm(P), x, y, x.copy(), nscal=1, visc=1/5000., schmidt=(1.0,), yuniform=True)
for k in range(3): d.substep_of_cycle(k, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(6): d.substep_of_cycle(3 + k, 1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per substep (all %d ranks): %.2f ms ; total %.2f ms" % (P, (t1 - t0) / 6 * 1e3, (t2 - t0) / 6 * 1e3))
