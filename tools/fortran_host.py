#!/usr/bin/env python3
"""The configuration north_star names -- the Fortran RK driver on the device path -- timed at the benchmark's size.

Writes tlab.ini, grid and restart files of an nx x ny x nz box (default 512^3, one scalar: BASELINE configs[2], bench.py's workload and fields)
in the reference's formats, runs the Fortran mini-driver (tlab_amd/fortran/test_rk_driver.f90: dns_main.f90's start-up + module TIME of
tools/dns/time.f90 on the drop-in modules) and reads `TIMING: ... ms_per_substep` from its tlab.log (wall clock between two tlab_sync around the
iterations after the warm-up ones, TLAB_AMD_TIMING).  Routes:

  unchanged            _build_rk: time.f90's statements as they are -- link-time RHS_GLOBAL_INCOMPRESSIBLE_1, then the reference's own DAXPY / DSCAL
                       calls; the library records them and runs ONE fused substep (csrc/deferred.cpp; the default of the Fortran host)
  unchanged_literal    the same executable with TLAB_AMD_DEFER=0: RHS, then 2 (3 + ns) BLAS passes over the fields, one by one
  fused                _build_rk_fused: the six-line patch of time.f90

Host layout of the arrays: q(isize_field, 3), s, hq, hs, txc(isize_txc_field, 9) as TLab_Initialize_Memory allocates them.  Prints one JSON line.
    python tools/fortran_host.py [--n 512] [--steps 6] [--warmup 1] [--routes unchanged,unchanged_literal,fused] [--repeat 1] [--keep DIR]
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EXE = {"unchanged": ("_build_rk", {}), "unchanged_literal": ("_build_rk", {"TLAB_AMD_DEFER": "0"}), "fused": ("_build_rk_fused", {})}

INI = """[Main]
Scalars=1
SpaceOrder1=CompactJacobian6
SpaceOrder2=CompactJacobian6Hyper

[Grid]
Imax={nx}
Jmax={ny}
Kmax={nz}
XUniform=yes
YUniform={yuni}
ZUniform=yes
XPeriodic=yes
YPeriodic=no
ZPeriodic=yes

[Parameters]
Reynolds=5000.0
Schmidt=1.0

[Time]
Scheme=RungeKuttaExplicit3
TimeStep={dtime}
Start=0
End={iters}

[BoundaryConditions]
VelocityJmin=noslip
VelocityJmax=noslip
Scalar1Jmin=dirichlet
Scalar1Jmax=dirichlet
"""


def write_case(d, nx, ny, nz, iters, dtime, stretch=False, seed=20250509):
    """bench.py's synthetic fields (smooth modes + 0.1 x uniform noise, walls at rest), generated on the host plane by plane"""
    from tlab_amd import io as tio
    x = np.arange(nx) / nx
    z = np.arange(nz) / nz
    y = np.arange(ny) / (ny - 1.0)
    if stretch:
        y = 0.5 * (1 + np.tanh(2.0 * (2 * y - 1)) / np.tanh(2.0))
    with open(os.path.join(d, "tlab.ini"), "w") as f:
        f.write(INI.format(nx=nx, ny=ny, nz=nz, iters=iters, dtime=repr(float(dtime)), yuni="no" if stretch else "yes"))
    tio.grid_write(os.path.join(d, "grid"), x, y, z, scales=[1.0, y[-1] - y[0], 1.0])
    rng = np.random.default_rng(seed)
    tp = 2 * np.pi
    X, Y = x[None, None, :], (np.arange(ny) / (ny - 1.0))[None, :, None]
    Z = z[:, None, None]
    wall = np.sin(np.pi * Y)
    shapes = [lambda: np.sin(tp * X) * np.cos(2 * tp * Y) * np.sin(3 * tp * Z), lambda: np.cos(tp * X) * np.sin(tp * Y) * np.sin(2 * tp * Z),
              lambda: np.sin(2 * tp * X) * np.cos(tp * Y) * np.cos(tp * Z), lambda: np.cos(3 * tp * X) * np.cos(tp * Y) * np.sin(tp * Z)]
    fields = []
    for sh in shapes:
        a = sh()
        a = (a + 0.1 * (2.0 * rng.random((nz, ny, nx)) - 1.0)) * wall
        fields.append(a.reshape(-1))
    tio.io_write_fields(os.path.join(d, "flow.0"), nx, ny, nz, 0, fields[:3], params=(0.0, 1.0 / 5000.0))
    tio.io_write_fields(os.path.join(d, "scal.0"), nx, ny, nz, 0, fields[3:], params=(0.0,))
    return fields


def run_route(d, route, warmup, extra_env=None, timeout=1800):
    sub, env = EXE[route]
    exe = os.path.join(ROOT, "tlab_amd", "fortran", sub, "test_rk_driver")
    if not os.path.exists(exe):
        return {"error": "%s not built" % os.path.relpath(exe, ROOT)}
    for name in ("tlab.log", "tlab.err"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            os.remove(p)
    t0 = time.time()
    r = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, TLAB_AMD_TIMING=str(warmup), **env, **(extra_env or {})))
    wall = time.time() - t0
    log = open(os.path.join(d, "tlab.log")).read() if os.path.exists(os.path.join(d, "tlab.log")) else ""
    if r.returncode != 0 or os.path.exists(os.path.join(d, "tlab.err")):
        err = open(os.path.join(d, "tlab.err")).read() if os.path.exists(os.path.join(d, "tlab.err")) else ""
        return {"error": (r.stdout + r.stderr + err + log)[-1500:]}
    m = re.search(r"TIMING: substeps\s+(\d+)\s+ms_per_substep\s+([0-9.Ee+-]+)", log)
    s = re.search(r"DEFERRED: [a-z_ ]+?((?:\s+\d+){6})\s*$", log, re.M)
    p = re.search(r"PLACEMENT: (.*)$", log, re.M)
    out = {"substeps": int(m.group(1)) if m else None, "ms_per_substep": float(m.group(2)) if m else None, "process_wall_s": round(wall, 1)}
    if s:
        c = [int(v) for v in s.group(1).split()]
        out["deferred"] = {"fused_substeps": c[0], "literal_flushes": c[1], "begin_steps": c[2], "eager_axpy": c[3], "eager_scal": c[4], "eager_zero": c[5]}
    if p:
        out["placement"] = p.group(1).strip()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--grid", type=int, nargs=3, default=None)
    ap.add_argument("--steps", type=int, default=6, help="timed Runge-Kutta steps (3 substeps each)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed Runge-Kutta steps in front")
    ap.add_argument("--routes", default="unchanged,unchanged_literal,fused")
    ap.add_argument("--repeat", type=int, default=1, help="processes per route (the arrays land elsewhere in every process)")
    ap.add_argument("--keep", default=None, help="directory to write the case into and keep (default: a temporary one)")
    ap.add_argument("--compare", action="store_true", help="also compare the final fields of the routes (bitwise: unchanged vs fused)")
    args = ap.parse_args()
    nx, ny, nz = args.grid or (args.n,) * 3
    d = args.keep or tempfile.mkdtemp(prefix="tlab_fortran_host_")
    os.makedirs(d, exist_ok=True)
    iters = args.warmup + args.steps
    t0 = time.time()
    write_case(d, nx, ny, nz, iters, 1e-3)
    out = {"what": "Fortran mini-driver (tlab_amd/fortran/test_rk_driver.f90: dns_main.f90 start-up + module TIME of time.f90 on the drop-in modules), host layout "
                   "q(isize_field,3) etc., restart files in the reference's format; ms per RK substep = wall clock between two tlab_sync around %d Runge-Kutta steps "
                   "after %d untimed; the plans are the HOST's (the reference's own FDM_Initialize, i.e. the wall closure 0.1 the flang build reads; bench.py's headline times "
                   "the consistent closure 0.0: same kernels, same bytes); TLab_AMD_Place_Arrays (6 candidates per block by default) ran at start-up" % (args.steps, args.warmup),
           "grid": [nx, ny, nz], "n_scalars": 1, "case_written_s": round(time.time() - t0, 1), "routes": {}}
    finals = {}
    for route in [r for r in args.routes.split(",") if r]:
        runs = [run_route(d, route, args.warmup) for _ in range(args.repeat)]
        ok = [r["ms_per_substep"] for r in runs if r.get("ms_per_substep")]
        out["routes"][route] = dict(runs[-1], ms_per_substep=min(ok) if ok else None, ms_all_processes=ok) if ok else runs[-1]
        if args.compare and ok:
            from tlab_amd import io as tio
            finals[route] = tio.io_read_fields(os.path.join(d, "flow.%d" % iters), nx, ny, nz, 3)[0] + tio.io_read_fields(os.path.join(d, "scal.%d" % iters), nx, ny, nz, 1)[0]
    if args.compare and "fused" in finals:
        for route, f in finals.items():
            if route != "fused":
                out["routes"][route]["fields_equal_fused_bitwise"] = bool(all(np.array_equal(a, b) for a, b in zip(f, finals["fused"])))
                out["routes"][route]["max_rel_diff_to_fused"] = float(max(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(f, finals["fused"])))
    if not args.keep:
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
