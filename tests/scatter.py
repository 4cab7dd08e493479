"""Conditioning of the composed paths, MEASURED on the oracle: how far do the oracle's own results move when its inputs are moved by
one unit in the last place?  The projection step of the pressure solver amplifies rounding (forcing div(q)/dte ~ 1e4-1e6 for a pressure of
1e2-1e3), so a composed-path tolerance cannot be a constant: the parity bound for such a path is

        device_err <= max(1e-12, 2 * scatter)

with `scatter` = max over a few one-ulp white-noise perturbations of the inputs of rel_err(oracle(perturbed), oracle(inputs)).  Where the
scatter is below 1e-12 the north-star tolerance applies unchanged.  TEST INFRASTRUCTURE (uses oracle/ only)."""
import numpy as np

FLOOR = 1e-12          # north_star: fp64 rel-err <= 1e-12


def one_ulp_noise(a, rng):
    """a moved by -1, 0 or +1 unit in the last place, independently per element (white noise of one ulp)."""
    a = np.asarray(a, dtype=np.float64)
    r = rng.integers(-1, 2, a.shape)
    up = np.nextafter(a, np.inf)
    dn = np.nextafter(a, -np.inf)
    return np.where(r > 0, up, np.where(r < 0, dn, a))


def _rel(a, b):
    s = np.abs(b).max()
    return float(np.abs(a - b).max() / (s if s > 0 else 1.0))


def scatter_of(fn, inputs, nsamples=2, seed=1234):
    """fn(*inputs) -> tuple of arrays.  Returns (base_outputs, [scatter per output]) with the inputs perturbed by one-ulp white noise."""
    rng = np.random.default_rng(seed)
    base = fn(*[np.array(a, copy=True) for a in inputs])
    sc = [0.0] * len(base)
    for _ in range(nsamples):
        out = fn(*[one_ulp_noise(a, rng) for a in inputs])
        sc = [max(s, _rel(o, b)) for s, o, b in zip(sc, out, base)]
    return base, sc


PARITY_RECORDS = []     # one dict per comparison against a bound(): written as profiles/<round>/parity_table.json at the end of a GPU session (conftest.py)


class Bound(float):
    """max(FLOOR, factor x scatter) that remembers where it came from.  `err <= bound(sc)` with a plain float on the left reaches __ge__ here first
    (the right operand is an instance of a subclass of float that overrides the reflected method), so every use of a scatter-derived bound is logged
    with the error it was compared against -- pytest -q prints none of them, the table keeps all of them."""

    def __new__(cls, scatter, factor, source="oracle one-ulp scatter", ref=None):
        b = super().__new__(cls, max(FLOOR, factor * scatter))
        b.scatter, b.factor, b.source = float(scatter), float(factor), source
        b.ref = ref          # (value, case[, one-ulp scatter of the reference's own routines]): by how much two builds of the REFERENCE differ on this very case
        # and field, and how far its own routines move under one ulp of input noise (tests/golden/yardsticks.json), or None
        return b

    def _log(self, err):
        import os
        PARITY_RECORDS.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "err": float(err), "yardstick": self.source,
                               "yardstick_value": self.scatter, "factor": self.factor, "bound": float(self), "ok": bool(float(err) <= float(self)),
                               "within_1e-12": bool(float(err) <= FLOOR),
                               "ref_build_diff": None if self.ref is None else float(self.ref[0]), "ref_build_case": None if self.ref is None else self.ref[1],
                               "ref_one_ulp_scatter": None if (self.ref is None or len(self.ref) < 3 or self.ref[2] is None) else float(self.ref[2]),
                               "bound_set_by": "north-star floor 1e-12" if float(self) <= FLOOR else "%g x %s" % (self.factor, self.source)})

    def __ge__(self, err):
        self._log(err)
        return float(self) >= float(err)

    def __gt__(self, err):
        self._log(err)
        return float(self) > float(err)

    __hash__ = float.__hash__


def bound(scatter, factor=2.0, source="oracle one-ulp scatter", ref=None):
    return Bound(scatter, factor, source, ref)


_YARD = None


def ref_yardstick(key):
    """tests/golden/yardsticks.json (made by tests/golden/make_golden_yardsticks.py in the build container): for the composed-path case `key` the
    relative difference between two builds of the REFERENCE'S OWN ROUTINES (amdflang -O2 with / without fused multiply-adds) composed into the same
    substeps on the same inputs (oracle/tlab_ref_rhs.py), per substep and field: {"q": [[..3..] per substep], "hq": ..., "s": ..., "hs": ...}; None
    when the table has no such case."""
    global _YARD
    if _YARD is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "yardsticks.json")
        _YARD = json.load(open(path))["cases"] if os.path.exists(path) else {}
    return _YARD.get(key)


def ref_of(key, k, name, i):
    """(value, case) for Bound(ref=...) or None"""
    y = ref_yardstick(key) if key else None
    if y is None or name not in y["diff"] or k >= len(y["diff"][name]) or i >= len(y["diff"][name][k]):
        return None
    u = y.get("ref_one_ulp_scatter")
    return (y["diff"][name][k][i], key, None if u is None else u[name][k][i])


CPU_PORT_SOURCE = "one-ulp scatter of the C/OpenMP port (oracle/tlab_cpu.c)"


def cpu_quota():
    """CPUs this process may use at once: the cgroup's CPU bandwidth quota if there is one, else the affinity mask (the GPU boxes show 256 logical CPUs and
    grant 16: an OpenMP team of 256 threads would only queue up)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return min(n, max(1, int(round(int(q) / int(per)))))
    except Exception:      # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return min(n, max(1, int(round(q / per))))
    except Exception:      # noqa: BLE001
        pass
    return n


_CPU_PORTS = {}


def cpu_port_factory(x, y, z, nscal, visc, schmidt, yuniform, hyper_bc1_ext=None):
    """make_oracle() for substep_scatter on the C + OpenMP restatement of the reference's CPU path (oracle/tlab_cpu.c through oracle/tlab_cpu.py) instead of the
    single-threaded numpy oracle: the checker for the cases with 1e7 points and more (the numpy oracle took 420 of the GPU suite's 735 s on
    four of them).  The port is held to the golden vectors of the reference and to the numpy oracle on the CPU (tests/test_cpu_baseline.py: operators
    <= 1e-14, substeps within the scatter bound); it restates no-slip walls / Dirichlet scalars with the default schemes only.  ONE driver per grid is
    built and handed out again with zeroed tendencies (its plan construction is numpy work of several seconds)."""
    from oracle import tlab_cpu as C
    key = (len(x), len(y), len(z), float(x[1] - x[0]), float(y[1] - y[0]), float(y[-1]), int(nscal), float(visc), tuple(float(v) for v in schmidt), bool(yuniform),
           hyper_bc1_ext)

    def make():
        L = C.load()
        L.tlabcpu_set_num_threads(max(1, min(cpu_quota(), L.tlabcpu_num_threads())))
        if key not in _CPU_PORTS:
            _CPU_PORTS.clear()                 # one grid at a time: a driver of 2048 x 1024 x 8 points with three scalars holds 3.5 GB
            _CPU_PORTS[key] = C.CpuDnsDriver(x, y, z, nscal=nscal, visc=visc, schmidt=tuple(schmidt), yuniform=yuniform, hyper_bc1_ext=hyper_bc1_ext)
        c = _CPU_PORTS[key]
        c.hq = [np.zeros(c.n) for _ in range(3)]
        c.hs = [np.zeros(c.n) for _ in range(nscal)]
        return c
    return make


_REF_FMA = None


def ref_build_diff(key):
    """By how much two legitimate builds of the REFERENCE (amdflang -O2 with and without fused multiply-adds, oracle/Makefile targets `all` and `fma`)
    differ on the case `key` of tests/golden/ref_fma_scatter.npz (made by tests/golden/make_golden_fma_scatter.py from oracle/_ref and oracle/_ref_fma)."""
    global _REF_FMA
    if _REF_FMA is None:
        import os
        _REF_FMA = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fma_scatter.npz"))
    return float(_REF_FMA["diff_" + key])


def ref_build_bound(key, factor=1.0):
    """max(1e-12, factor x the difference between two builds of the reference itself): the device may differ from the oracle by what the reference
    differs from itself when its compiler fuses multiply-adds -- a yardstick that does not come from this repository's oracle."""
    return Bound(ref_build_diff(key), factor, "difference between the reference's own FMA / non-FMA builds (%s)" % key)


def substep_scatter(make_oracle, q0, s0, schedule, nsamples=1, seed=77):
    """Oracle run of the substeps in `schedule` = [(dte, kco, scale[, new_step]), ...] from the fields q0 (3) + s0 (nscal), plus `nsamples` runs
    from one-ulp perturbations of those fields; new_step: the tendencies are zeroed before that substep (TIME_RUNGEKUTTA, time.f90:212-216).  Returns (base, scat): per substep a dict name -> list of arrays / of scatters for
    name in q, hq, s, hs."""
    ns = len(s0)

    def run(*fields):
        o = make_oracle()
        for i in range(3):
            o.q[i] = np.array(fields[i], copy=True)
        for i in range(ns):
            o.s[i] = np.array(fields[3 + i], copy=True)
        outs = []
        for item in schedule:
            dte, kco, scale = item[:3]
            if len(item) > 3 and item[3]:
                o.hq = [np.zeros_like(a) for a in o.hq]
                o.hs = [np.zeros_like(a) for a in o.hs]
            o.time_substep(dte, kco, scale)
            outs += [a.copy() for a in o.q + o.hq + o.s + o.hs]
        return tuple(outs)

    base, sc = scatter_of(run, list(q0) + list(s0), nsamples, seed)
    per = 6 + 2 * ns
    B, S = [], []
    for k in range(len(schedule)):
        b, s = base[k * per:(k + 1) * per], sc[k * per:(k + 1) * per]
        B.append({"q": b[0:3], "hq": b[3:6], "s": b[6:6 + ns], "hs": b[6 + ns:6 + 2 * ns]})
        S.append({"q": s[0:3], "hq": s[3:6], "s": s[6:6 + ns], "hs": s[6 + ns:6 + 2 * ns]})
    return B, S
