import os
import sys
import glob
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_extra: GPU tests of features OUTSIDE SURVEY.md section 8 (staggered pressure, anelastic weights): built in "
                                       "earlier rounds, kept working, not part of the hot-path parity evidence.  Run with -m gpu_extra; -m gpu leaves them out "
                                       "except for a smoke subset (two staggered substeps, the anelastic Burgers operators); OPR_Helmholtz (section 8f n3) is under -m gpu")


# The driver runs `pytest tests/ -x -q -m gpu` under a wall-clock limit (round 4: killed at 1200 s after 294 of 398 tests).  Parity-critical tests
# first, in the order of SURVEY.md section 8: the operators (a1-a9), the Poisson solver (a10-a13), the RHS / RK substep (n1, n2), the BASELINE
# configs at size, the drop-in boundary (b), the decomposed drivers (e, a14), then the non-default schemes and formats (n3, n4).
GPU_ORDER = ["test_gpu_derivs", "test_gpu_poisson.py", "test_gpu_rhs", "test_gpu_deferred", "test_gpu_configs", "test_gpu_fortran_dropin", "test_gpu_valid_programs", "test_gpu_slab_native",
             "test_gpu_slab.py", "test_gpu_native_trp", "test_gpu_pencil", "test_gpu_dist", "test_io_formats", "test_gpu_poisson_direct", "test_direct_scheme",
             "test_gpu_filter", "test_gpu_placement", "test_stagger"]


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    expr = config.getoption("markexpr", "") or ""
    if "gpu_extra" not in expr:
        keep, drop = [], []
        for it in items:
            (drop if it.get_closest_marker("gpu_extra") is not None else keep).append(it)
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep

    def rank(it):
        if it.get_closest_marker("gpu") is None and it.get_closest_marker("gpu_extra") is None:
            return -1                      # CPU tests keep their place in front
        for i, name in enumerate(GPU_ORDER):
            if name in it.nodeid:
                return i
        return len(GPU_ORDER)
    items.sort(key=rank)                   # stable: the order inside a file is the file's


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def rel_err(a, b):
    """max|a-b| / max|b| (the fp64 parity measure; tolerance per north_star is 1e-12)."""
    a = np.asarray(a)
    b = np.asarray(b)
    s = np.abs(b).max()
    return float(np.abs(a - b).max() / (s if s > 0 else 1.0))


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


PARITY_ROUND = "r06"


def pytest_sessionfinish(session, exitstatus):
    """Composed-path parity, auditable: every comparison of a device (or CPU-port) error against a scatter-derived bound made in this session goes
    to profiles/<round>/parity_table.json -- per test the measured error, the oracle's own one-ulp scatter, the bound used and what set it.  Written
    on GPU sessions only (the -m "not gpu" suite holds two such comparisons); a copy under gpurun_out/ travels back from the GPU box."""
    try:
        import json
        import subprocess
        import torch
        from scatter import PARITY_RECORDS
        if not PARITY_RECORDS or not torch.cuda.is_available():
            return
        rows = {}
        for r in PARITY_RECORDS:          # worst comparison per (test, bound): a test checks many fields against one scatter list
            key = (r["test"], r["bound_set_by"])
            k = rows.setdefault(key, dict(r, comparisons=0, max_err=0.0, max_err_over_bound=0.0, max_yardstick=0.0, max_bound=0.0, max_ref=None, ref_case=None))
            if r.get("ref_build_diff") is not None:
                k["max_ref"] = max(k["max_ref"] or 0.0, r["ref_build_diff"])
                k["ref_case"] = r["ref_build_case"]
            if r.get("ref_one_ulp_scatter") is not None:
                k["max_ref_ulp"] = max(k.get("max_ref_ulp") or 0.0, r["ref_one_ulp_scatter"])
            k["comparisons"] += 1
            k["max_err"] = max(k["max_err"], r["err"])
            k["max_err_over_bound"] = max(k["max_err_over_bound"], r["err"] / r["bound"])
            k["max_yardstick"] = max(k["max_yardstick"], r["yardstick_value"])
            k["max_bound"] = max(k["max_bound"], r["bound"])
            k["ok"] = k["ok"] and r["ok"]
        table = [{"test": t, "bound_set_by": by, "yardstick": k["yardstick"], "comparisons": k["comparisons"], "max_err": k["max_err"],
                  "max_yardstick": k["max_yardstick"], "max_bound": k["max_bound"], "max_err_over_bound": k["max_err_over_bound"], "all_within": k["ok"],
                  "within_1e-12": bool(k["max_err"] <= 1e-12),
                  # the reference against itself on this very case (two builds of its own routines composed into the same substeps): tests/golden/yardsticks.json
                  "ref_build_diff": k["max_ref"], "ref_build_case": k["ref_case"], "ref_one_ulp_scatter": k.get("max_ref_ulp"),
                  "err_over_ref_build": (k["max_err"] / k["max_ref"]) if k["max_ref"] else None} for (t, by), k in sorted(rows.items())]
        try:
            head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
            head = head or os.environ.get("TLAB_COMMIT") or None        # the GPU box has no .git: the caller may pass the commit ...
            stamp = os.path.join(ROOT, "tlab_amd", "BUILD_COMMIT")       # ... or build() left it beside the libraries it built (__graft_entry__.py)
            if head is None and os.path.exists(stamp):
                head = open(stamp).read().strip() or None
        except Exception:       # noqa: BLE001
            head = None
        doc = {"what": "device error vs yardstick vs bound for every composed-path comparison of this pytest session (tests/scatter.py::Bound); yardstick = "
                       "the oracle's own one-ulp scatter, or the difference between two builds of the reference itself (tests/golden/ref_fma_scatter.npz)",
               "bound": "max(1e-12, factor x yardstick)", "commit": head, "exitstatus": int(exitstatus),
               "rows_above_floor": sum(1 for r in table if r["max_bound"] > 1e-12),
               "rows_with_error_above_1e-12": sum(1 for r in table if not r["within_1e-12"]),
               "rows_above_1e-12_without_a_reference_made_figure": [r["test"] for r in table if (not r["within_1e-12"] or r["max_bound"] > 1e-12) and r["ref_build_diff"] is None
                                                                   and "difference between the reference" not in r["yardstick"]],
               "rows": table}
        # a run of part of the suite must not replace the table of a whole one
        fname = "parity_table.json" if getattr(session, "testscollected", 0) >= 300 else "parity_table_partial_run.json"
        doc["tests_in_session"] = int(getattr(session, "testscollected", 0))
        for d in (os.path.join(ROOT, "profiles", PARITY_ROUND), os.path.join(ROOT, "gpurun_out", PARITY_ROUND)):
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, fname), "w") as f:
                json.dump(doc, f, indent=1)
    except Exception as e:       # noqa: BLE001  (bookkeeping must never fail a test session)
        print("parity table not written:", e)
