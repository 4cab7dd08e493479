import sys, numpy as np, torch
sys.path.insert(0, ".")
import tlab_amd as T
T.init(0)
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[iters // 2]
for nx, ny, nz in ((2048, 1024, 64), (1024, 512, 128)):
    x = np.arange(nx) / nx; z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(2))
    g = [T.FdmPlan(x, True, True), T.FdmPlan(y, False, False), T.FdmPlan(z, True, True)]
    N = nx * ny * nz
    u = torch.rand(N, dtype=torch.float64, device="cuda") - 0.5; v = torch.rand(N, dtype=torch.float64, device="cuda") - 0.5
    r = torch.empty_like(u); t = torch.empty_like(u)
    for d, part, burg in ((1, T.OPR_Partial_X, T.OPR_Burgers_X), (2, T.OPR_Partial_Y, T.OPR_Burgers_Y), (3, T.OPR_Partial_Z, T.OPR_Burgers_Z)):
        tp = timeit(lambda: part(T.OPR_P1, nx, ny, nz, 0, g[d - 1], u, r, t)); path1 = T.load().tlab_last_kernel_path()
        tb = timeit(lambda: burg(T.OPR_B_U_IN, 1e-3, nx, ny, nz, 0, g[d - 1], u, v, r, t)); path2 = T.load().tlab_last_kernel_path()
        print((nx, ny, nz), "dir %d: P1 %.0f GB/s (path %d)  Burgers %.0f GB/s (path %d)" % (d, 16 * N / tp / 1e6, path1, 24 * N / tb / 1e6, path2), flush=True)
