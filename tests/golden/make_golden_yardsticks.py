#!/usr/bin/env python3
"""The reference against itself on the COMPOSED path: every composed-path parity case of tests/cases.py -- RK substeps of the
whole hot path from the very fields, grids and schedules the GPU tests use -- is run through TWO legitimate builds of the reference's own routines,
composed by oracle/tlab_ref_rhs.py::RefComposedDns (FDM_Der1/2_Solve, the Burgers composition, FDM_Int1 + OPR_ODE2_Factorize_NN per Fourier mode,
BOUNDARY_BCS_NEUMANN_Y: all compiled from /root/reference where it lies; pointwise sums and FFTs in numpy on both sides):

    make -C oracle        -> oracle/_ref      amdflang -O2, x86-64 baseline (no fused multiply-add exists)
    make -C oracle fma    -> oracle/_ref_fma  amdflang -O2 -march=haswell (a*b+c contracted into vfmadd)

What is stored (tests/golden/yardsticks.json): per case, per substep and field (q, hq, s, hs) the relative difference max|fma - plain| / max|plain|, plus
the difference of the plain build from the numpy oracle on the same case (the oracle's distance from the reference, composed path included).  The GPU
tests print these figures next to every device error that is bounded above 1e-12 (tests/scatter.py::ref_of, profiles/<round>/parity_table.json column
ref_build_diff).  Nothing here is a device result.  One subprocess per (case, build): the reference keeps its plans in module variables.

    python tests/golden/make_golden_yardsticks.py [--only substring] [--jobs N]      (needs /root/reference; builds both libraries if missing)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIBS = {"plain": os.path.join(ROOT, "oracle", "_ref", "libtlab_ref.so"), "fma": os.path.join(ROOT, "oracle", "_ref_fma", "libtlab_ref.so")}
OUT = os.path.join(ROOT, "tests", "golden", "yardsticks.json")
NAMES = ("q", "hq", "s", "hs")


def run_case(case, make, nsamples=0):
    from scatter import substep_scatter
    return substep_scatter(make, case["q0"], case["s0"], case["sched"], nsamples=nsamples)


def worker(index, out, with_numpy):
    import cases as C
    fn, a, k = C.registry()[index]
    case = fn(*a, **k)
    from oracle.tlab_ref_rhs import RefComposedDns
    # plain build: also the reference's OWN conditioning -- its routines re-run from fields moved by one ulp of white noise (tests/scatter.py, same seeds as
    # the tests use for the numpy oracle's scatter)
    B, S = run_case(case, C.make_oracle_factory(case, RefComposedDns), nsamples=2 if with_numpy is not None else 0)
    res = {"%s_%d_%d" % (name, kk, i): arr for kk, b in enumerate(B) for name in NAMES for i, arr in enumerate(b[name])}
    if with_numpy is not None:
        for kk, sc in enumerate(S):
            for name in NAMES:
                for i, v in enumerate(sc[name]):
                    res["ulp_%s_%d_%d" % (name, kk, i)] = np.array(float(v))
    if with_numpy:      # the numpy oracle on the same case, compared in this process (no second copy of the fields on disk)
        Bn, _ = run_case(case, C.make_oracle_factory(case))
        for kk, b in enumerate(Bn):
            for name in NAMES:
                for i, arr in enumerate(b[name]):
                    ref = res["%s_%d_%d" % (name, kk, i)]
                    s = np.abs(ref).max()
                    res["numpy_%s_%d_%d" % (name, kk, i)] = np.array(float(np.abs(arr - ref).max() / (s if s > 0 else 1.0)))
    np.savez(out, **res)


def worker_direct(index, out, with_ulp):
    """dp/dy = OPR_Partial_Y(p) of the direct elliptic solver (opr_elliptic.f90:447-449) by the reference's own OPR_Partial_Y on the pressure of the numpy
    oracle (bitwise the reference's per-mode arithmetic, FFTs by numpy): the two builds on the same p, and the plain build on p moved by one ulp"""
    import cases as C
    from oracle import ref_lib as R, tlab_oracle as O, tlab_oracle_poisson as OP
    from scatter import one_ulp_noise
    nx, ny, nz, ibc = C.registry_direct()[index]
    case = C.poisson_direct(nx, ny, nz, ibc)
    ogx, ogz = O.FdmPlan(case["x"], True, True), O.FdmPlan(case["z"], True, True) if nz > 1 else None
    ogy = O.FdmPlan.from_tables(case["tab"], mode2=case["mode2"])
    oplan = OP.PoissonDirectPlan(ogx, ogy, ogz if nz > 1 else ogx, nx, ny, nz)
    p_ref, _ = OP.opr_poisson_fxz_direct(oplan, case["f"], case["hb"], case["ht"], ibc)
    R.init(nx, ny, nz)
    R.fdm_create(1, case["x"], True, True, 6, 16)
    R.fdm_create(2, case["y"], False, False, 6, 16)
    if nz > 1:
        R.fdm_create(3, case["z"], True, True, 6, 16)
    res = {"dpdy": R.partial(2, 1, nx, ny, nz, 0, p_ref)[0]}
    if with_ulp:
        rng = np.random.default_rng(1234)
        sc = 0.0
        for _ in range(2):
            d = R.partial(2, 1, nx, ny, nz, 0, one_ulp_noise(p_ref, rng))[0]
            sc = max(sc, float(np.abs(d - res["dpdy"]).max() / np.abs(res["dpdy"]).max()))
        res["ulp"] = np.array(sc)
    np.savez(out, **res)


def rel(a, b):
    s = np.abs(b).max()
    return float(np.abs(a - b).max() / (s if s > 0 else 1.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--numpy-limit", type=int, default=3_000_000, help="cases up to this many points also record the numpy oracle's distance from the plain build")
    args = ap.parse_args()
    if not all(os.path.exists(p) for p in LIBS.values()):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "fma"], check=True)
    import cases as C
    table = json.load(open(OUT))["cases"] if os.path.exists(OUT) else {}
    tmp = tempfile.mkdtemp()
    for index, (fn, a, k) in enumerate(C.registry()):
        case = fn(*a, **k)
        key = case["key"]
        if args.only and args.only not in key:
            continue
        npts = len(case["x"]) * len(case["y"]) * len(case["z"])
        t0 = time.time()
        outs = {}
        for tag, lib in LIBS.items():
            outs[tag] = os.path.join(tmp, "%d_%s.npz" % (index, tag))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(index), outs[tag],
                            "none" if tag != "plain" else ("1" if npts <= args.numpy_limit else "0")],
                           check=True, env=dict(os.environ, TLAB_REF_LIB=lib))
        A, B = np.load(outs["fma"]), np.load(outs["plain"])
        nsub = len(case["sched"])
        counts = {"q": 3, "hq": 3, "s": case["nscal"], "hs": case["nscal"]}
        diff = {name: [[rel(A["%s_%d_%d" % (name, kk, i)], B["%s_%d_%d" % (name, kk, i)]) for i in range(counts[name])] for kk in range(nsub)] for name in NAMES}
        entry = {"grid": [len(case["x"]), len(case["y"]), len(case["z"])], "nscal": case["nscal"], "walls": case["walls"], "substeps": nsub, "diff": diff}
        entry["ref_one_ulp_scatter"] = {name: [[float(B["ulp_%s_%d_%d" % (name, kk, i)]) for i in range(counts[name])] for kk in range(nsub)] for name in NAMES}
        if "numpy_q_0_0" in B.files:
            entry["numpy_oracle_vs_plain_build"] = {name: [[float(B["numpy_%s_%d_%d" % (name, kk, i)]) for i in range(counts[name])] for kk in range(nsub)] for name in NAMES}
        table[key] = entry
        worst = max(v for name in NAMES for row in diff[name] for v in row) if nsub else 0.0
        print("%-48s %9d points  worst ref-vs-ref %.2e  (%.0f s)" % (key, npts, worst, time.time() - t0), flush=True)
        for f in outs.values():
            os.remove(f)
        ver = subprocess.run(["amdflang", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
        doc = {"what": "max|fma build - plain build| / max|plain build| of the reference's own routines composed into the RK substeps of each case (oracle/tlab_ref_rhs.py), "
                       "per substep and field; ref_one_ulp_scatter: the plain build re-run from fields moved by one ulp of white noise (the reference's own conditioning, 2 samples); "
                       "numpy_oracle_vs_plain_build: the numpy oracle's distance from the plain build on the same case",
               "compiler": ver, "plain": "-O2 (x86-64 baseline: no fused multiply-add)", "fma": "-O2 -march=haswell (vfmadd contraction)",
               "made_by": "tests/golden/make_golden_yardsticks.py", "cases": table}
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    for index, (nx, ny, nz, ibc) in enumerate(C.registry_direct()):
        key = "poisson_direct.dpdy[%d-%d-%d-%d]" % (nx, ny, nz, ibc)
        if args.only and args.only not in key:
            continue
        outs = {}
        for tag, lib in LIBS.items():
            outs[tag] = os.path.join(tmp, "d%d_%s.npz" % (index, tag))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--worker-direct", str(index), outs[tag], "1" if tag == "plain" else "0"], check=True,
                           env=dict(os.environ, TLAB_REF_LIB=lib))
        A, B = np.load(outs["fma"]), np.load(outs["plain"])
        table[key] = {"grid": [nx, ny, nz], "substeps": 1, "diff": {"dpdy": [[rel(A["dpdy"], B["dpdy"])]]}, "ref_one_ulp_scatter": {"dpdy": [[float(B["ulp"])]]},
                      "what": "OPR_Partial_Y of the reference on the oracle's pressure: two builds on the same p / the plain build on p moved by one ulp"}
        print("%-48s ref-vs-ref %.2e  one-ulp %.2e" % (key, table[key]["diff"]["dpdy"][0][0], float(B["ulp"])), flush=True)
        doc = json.load(open(OUT))
        doc["cases"] = table
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[1] == "--worker-direct":
        worker_direct(int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1")
    elif len(sys.argv) > 4 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), sys.argv[3], None if sys.argv[4] == "none" else sys.argv[4] == "1")
    else:
        main()
