"""Inputs of the composed-path parity cases (RK substeps of the whole hot path) in ONE place: the GPU tests build their fields, grids and schedules
here, and tests/golden/make_golden_yardsticks.py runs the very same cases through two builds of the reference's own routines in the build container
(oracle/tlab_ref_rhs.py) to make the reference-vs-reference figure each parity row is printed next to (tests/golden/yardsticks.json).

A case = dict(key, x, y, z, nscal, visc, sc, yuniform, walls, q0, s0, sched): walls = None (no-slip, Dirichlet scalars) or (VelocityJmin, VelocityJmax,
ScalarJmin, ScalarJmax) keywords of [BoundaryConditions]; sched = [(dte, kco, scale[, new_step]), ...] as tests/scatter.py::substep_scatter reads it.
TEST INFRASTRUCTURE (numpy only)."""
import numpy as np

KDT = [1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0]          # RungeKuttaExplicit3 (tools/dns/time.f90:100-107)
KCO = [-5.0 / 9.0, -153.0 / 128.0]
DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN = 3, 4          # tools/dns/boundary_bcs.f90:18-19


def grids(nx, ny, nz, stretch):
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz * 1.0
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5)) if stretch else np.arange(ny) / (ny - 1.0)
    return x, y, z


def init_fields(nx, ny, nz, x, y, z, seed, noise=0.1):
    rng = np.random.default_rng(seed)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))                  # vanishes on the walls (no-slip)
    u = (np.sin(np.pi * X) * np.cos(2 * np.pi * Z) + noise * rng.uniform(-1, 1, X.shape)) * wall
    v = (np.cos(np.pi * X) * np.sin(2 * np.pi * Z) + noise * rng.uniform(-1, 1, X.shape)) * wall ** 2
    w = (np.sin(2 * np.pi * X + 1) * np.sin(2 * np.pi * Z) + noise * rng.uniform(-1, 1, X.shape)) * wall
    s = np.cos(np.pi * X) * Y + noise * rng.uniform(-1, 1, X.shape)
    return [a.ravel() for a in (u, v, w)], [s.ravel()]


def slab_grid(nx, ny, nz):
    x = np.arange(nx) / nx * 2.0
    z = np.arange(nz) / nz
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    return x, y, z


def slab_fields(x, y, z, seed, count=4):
    rng = np.random.default_rng(seed)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    return [((np.sin(np.pi * X + k) * np.cos(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(count)]


def neumann_fields(x, y, z, seed):
    rng = np.random.default_rng(seed)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * Y)
    q0 = [(np.sin(np.pi * X) * np.cos(2 * np.pi * Z) * np.cos(np.pi * Y) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel(),
          ((np.cos(np.pi * X) * np.sin(2 * np.pi * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel(),
          (np.sin(2 * np.pi * X + 1) * np.sin(2 * np.pi * Z) * np.cos(np.pi * Y) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()]
    s0 = [(np.cos(np.pi * X) * np.cos(np.pi * Y) + 0.1 * rng.uniform(-1, 1, X.shape)).ravel()]
    return q0, s0


def rk_sched(dtime, nsub=2):
    return [(dtime * KDT[k], KCO[k], True) for k in range(nsub)]


# ---- tests/test_gpu_rhs.py ----
def rhs_substep(nx, ny, nz, stretch, hyper=None):
    """hyper: None = the wall closure the reference as compiled reads (0.1); 0.0 = the consistent closure bench.py times -- its reference-made figures
    come from the reference's own routines on ITS plan with that one entry replaced (oracle/ref_driver.f90::ref_fdm_set_hyper_bc1_ext)"""
    x, y, z = grids(nx, ny, nz, stretch)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 3)
    key = "rhs.substep[%d-%d-%d-%s]" % (nx, ny, nz, stretch) if hyper is None else "rhs.substep[%d-%d-%d-%s-closure%g]" % (nx, ny, nz, stretch, hyper)
    return dict(key=key, x=x, y=y, z=z, nscal=1, visc=1.0 / 800.0, sc=(0.7,), yuniform=not stretch, walls=None, hyper=hyper,
                q0=q0, s0=s0, sched=rk_sched(2e-3))


def rhs_neumann(vel, scal, nx=64, seed=11):
    """free-slip walls / Neumann scalars on 64 x 64 x 32 (test_substep_with_neumann_walls_vs_oracle) and 256 x 64 x 64 (test_neumann_wall_planes_route)"""
    ny, nz = (64, 32) if nx == 64 else (64, 64)
    x, y, z = grids(nx, ny, nz, True)
    q0, s0 = neumann_fields(x, y, z, seed)
    return dict(key="rhs.neumann[%d-%s-%s-%s-%s]" % (nx, vel[0], vel[1], scal[0], scal[1]), x=x, y=y, z=z, nscal=1, visc=1.0 / 800.0, sc=(0.7,), yuniform=False,
                walls=(vel[0], vel[1], scal[0], scal[1]), q0=q0, s0=s0, sched=rk_sched(2e-3))


def rhs_lines(nx, ny, nz, nscal, stretch):
    """the line lengths of BASELINE configs[3] / [4] with the other extents reduced (test_line_lengths_of_the_large_configs)"""
    x, y, z = grids(nx, ny, nz, stretch)
    q0, s0 = init_fields(nx, ny, nz, x, y, z, 23, noise=1e-3)
    ss = [s0[0] * (1.0 + 0.3 * i) + 0.1 * i for i in range(nscal)]
    return dict(key="rhs.lines[%d-%d-%d-%d-%s]" % (nx, ny, nz, nscal, stretch), x=x, y=y, z=z, nscal=nscal, visc=1.0 / 5000.0, sc=(0.7, 1.0, 2.5)[:nscal],
                yuniform=not stretch, walls=None, q0=q0, s0=ss, sched=rk_sched(1e-3))


# ---- tests/test_gpu_slab.py, test_gpu_slab_native.py, test_gpu_pencil.py: decomposed drivers against the single domain AND the oracle ----
def slab(P, nx, ny, nz, bcs):
    x, y, z = slab_grid(nx, ny, nz)
    f = slab_fields(x, y, z, P)
    return dict(key="slab[%d-%d-%d-%d-%s]" % (P, nx, ny, nz, bcs), x=x, y=y, z=z, nscal=1, visc=1.0 / 600.0, sc=(0.8,), yuniform=False,
                walls=("freeslip", "freeslip", "neumann", "dirichlet") if bcs == "freeslip" else None, q0=f[:3], s0=f[3:4], sched=rk_sched(2e-3))


def slab_native(bcs):
    nx, ny, nz = 128, 24, 256
    x, y, z = slab_grid(nx, ny, nz)
    f = slab_fields(x, y, z, 11, count=5)
    return dict(key="slab_native[%s]" % bcs, x=x, y=y, z=z, nscal=1, visc=1.0 / 600.0, sc=(0.8,), yuniform=False,
                walls=("freeslip", "freeslip", "neumann", "dirichlet") if bcs == "freeslip" else None, q0=f[:3], s0=f[3:4], sched=rk_sched(2e-3))


def pencil(npi, npk, nx, ny, nz, bcs):
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * np.pi
    y = 0.5 * (1 + np.tanh(1.5 * (2 * np.arange(ny) / (ny - 1) - 1)) / np.tanh(1.5))
    rng = np.random.default_rng(10 * npi + npk)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    wall = np.sin(np.pi * (Y - y[0]) / (y[-1] - y[0]))
    f = [((np.sin(X + k) * np.cos(2 * Z) + 0.1 * rng.uniform(-1, 1, X.shape)) * wall).ravel() for k in range(4)]
    return dict(key="pencil[%d-%d-%d-%d-%d-%s]" % (npi, npk, nx, ny, nz, bcs), x=x, y=y, z=z, nscal=1, visc=1.0 / 300.0, sc=(0.7,), yuniform=False,
                walls=("freeslip", "freeslip", "neumann", "dirichlet") if bcs == "freeslip" else None, q0=f[:3], s0=f[3:4], sched=rk_sched(2e-3))


# ---- tests/test_gpu_poisson_direct.py: EllipticOrder = CompactDirect6; dp/dy = OPR_Partial_Y(p) differentiates the rounding noise of p ----
def poisson_direct(nx, ny, nz, ibc):
    import glob
    import os
    g = np.load(sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct_y*.npz")))[0])
    tab = {k[len("ny%d_" % ny):]: g[k] for k in g.files if k.startswith("ny%d_" % ny)}
    y = tab["nodes"]
    x = np.arange(nx) / nx * 2 * np.pi
    z = np.arange(nz) / nz * np.pi if nz > 1 else np.zeros(1)
    rng = np.random.default_rng(ny + ibc)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    f = (np.sin(X) * np.cos(2 * Z) * np.exp(Y) + 0.3 * rng.uniform(-1, 1, X.shape)).ravel()
    hb, ht = rng.uniform(-1, 1, (nz, nx)), rng.uniform(-1, 1, (nz, nx))
    return dict(key="poisson_direct.dpdy[%d-%d-%d-%d]" % (nx, ny, nz, ibc), x=x, y=y, z=z, tab=tab, mode2=16, ibc=ibc, f=f, hb=hb, ht=ht)


def registry_direct():
    return [(16, 64, 1, 3), (32, 128, 8, 3), (64, 512, 16, 3)]


def make_oracle_factory(case, cls=None):
    """() -> a fresh oracle of the case (numpy DnsOracle, or another class with its interface) with the case's walls"""
    from oracle.tlab_oracle_rhs import DnsOracle
    cls = cls or DnsOracle

    def vel(kind):
        return [DNS_BCS_DIRICHLET] * 3 if kind == "noslip" else [DNS_BCS_NEUMANN, DNS_BCS_DIRICHLET, DNS_BCS_NEUMANN]

    def make():
        kw = {} if case.get("hyper") is None else {"hyper_bc1_ext": case["hyper"]}
        o = cls(case["x"], case["y"], case["z"], nscal=case["nscal"], visc=case["visc"], schmidt=case["sc"], yuniform=case["yuniform"], **kw)
        if case["walls"]:
            w = case["walls"]
            o.flow_jmin, o.flow_jmax = vel(w[0]), vel(w[1])
            o.scal_jmin = [DNS_BCS_NEUMANN if w[2] == "neumann" else DNS_BCS_DIRICHLET] * case["nscal"]
            o.scal_jmax = [DNS_BCS_NEUMANN if w[3] == "neumann" else DNS_BCS_DIRICHLET] * case["nscal"]
        return o
    return make


def registry():
    """key -> () -> case, for every composed-path case whose bound may exceed 1e-12 (first substeps from a non-solenoidal field)"""
    R = {}

    def add(fn, *a, **k):
        probe_key = fn.__name__, a, tuple(sorted(k.items()))
        R[probe_key] = (fn, a, k)
    for g in [(32, 40, 16, True), (64, 32, 32, False), (256, 64, 64, True)]:
        add(rhs_substep, *g)
    for g in [(32, 40, 16, True), (64, 32, 32, False), (256, 64, 64, True)]:      # the same from the consistent wall closure (what bench.py times)
        add(rhs_substep, *g, hyper=0.0)
    for vel, scal in [(("freeslip", "freeslip"), ("neumann", "dirichlet")), (("noslip", "freeslip"), ("dirichlet", "neumann"))]:
        add(rhs_neumann, vel, scal)
    for vel, scal in [(("freeslip", "freeslip"), ("neumann", "neumann")), (("freeslip", "noslip"), ("dirichlet", "neumann"))]:
        add(rhs_neumann, vel, scal, nx=256, seed=21)
    for g in [(1024, 512, 16, 1, False), (2048, 1024, 8, 3, True), (16, 32, 2048, 1, False), (32, 16, 1024, 1, False)]:
        add(rhs_lines, *g)
    for g in [(2, 64, 32, 32, "noslip"), (4, 32, 64, 64, "noslip"), (8, 64, 32, 64, "noslip"), (4, 64, 32, 32, "freeslip"), (2, 32, 16, 128, "noslip"),
              (4, 64, 24, 256, "freeslip"), (8, 32, 16, 512, "noslip"), (3, 48, 16, 192, "noslip")]:
        add(slab, *g)
    for b in ("noslip", "freeslip"):
        add(slab_native, b)
    for g in [(2, 2, 32, 24, 16, "noslip"), (2, 4, 32, 16, 32, "noslip"), (4, 2, 64, 16, 16, "freeslip"), (2, 1, 32, 24, 8, "noslip"), (8, 1, 64, 8, 16, "noslip"),
              (1, 4, 32, 16, 16, "noslip")]:
        add(pencil, *g)
    return [(fn, a, k) for fn, a, k in R.values()]
