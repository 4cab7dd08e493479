"""OPR_Partial_{X,Y,Z}(OPR_P1) at 256^3 with the tridiagonal (CompactJacobian6) and the pentadiagonal (CompactJacobian6Penta) first derivative:
ms, algorithmic GB/s and the kernel path taken (1 generic / k_penta1, 2 wave-per-line, 3 register tile).  DESIGN.md section 8."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import tlab_amd as T
T.init(0)
import json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = np.arange(n) / n
y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(n) / (n - 1) - 1)) / np.tanh(1.5))
u = torch.rand(n ** 3, dtype=torch.float64, device="cuda")
r = torch.empty_like(u); t = torch.empty_like(u)
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, s1 in (("jacobian6", 4), ("penta", 5)):
    for d, (nodes, per) in {1: (x, True), 2: (y, False), 3: (x, True)}.items():
        try:
            g = T.FdmPlan(nodes, per, per, s1, 7) if name == "penta" else T.FdmPlan(nodes, per, per)
            f = (T.OPR_Partial_X, T.OPR_Partial_Y, T.OPR_Partial_Z)[d - 1]
            ms = timeit(lambda: f(T.OPR_P1, n, n, n, 0, g, u, r, t))
            print(json.dumps({"scheme": name, "n": n, "dir": d, "op": "OPR_Partial(P1)", "ms": round(ms, 4), "GBps": round(16 * n ** 3 / ms / 1e6, 1),
                              "path": T.load().tlab_last_kernel_path(), "x_tile": os.environ.get("TLAB_PENTA_TILE_X", "1")}), flush=True)
        except Exception as e:
            print(name, d, "ERR", e)
