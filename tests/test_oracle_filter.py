"""The oracle's restatement of the 1-D filters behind [Dealiasing] (oracle/tlab_oracle_filter.py) against vectors the reference's own filter
modules produced (tests/golden/filters.npz): bitwise, for compact / explicit6 / explicit4 / compactcutoff, periodic and biased / free / zero ends;
and the dealiasing branch of the Burgers oracle against its definition."""
import os
import re
import numpy as np
import pytest

from oracle import tlab_oracle as O
from oracle import tlab_oracle_filter as F

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filters.npz"))


def filter_of(key):
    n, t, per, b0, b1 = (int(v) for v in re.match(r"n(\d+)_t(\d+)_p(\d)_b(\d)(\d)", key).groups())
    c = G[key + "_coeffs"]
    return F.Filter(t, n, bool(per), c if c.shape[1] else None, b0, b1), n


@pytest.mark.parametrize("key", [str(k) for k in G["cases"]])
def test_filters_match_the_reference_bitwise(key):
    f, n = filter_of(key)
    assert np.array_equal(F.opr_filter_1d(f, G["n%d_u" % n]), G[key + "_res"])


def test_dealiased_burgers_is_the_filtered_product():
    """OPR_Burgers_1D with dealiasing (opr_burgers.f90:478-500): result = nu d2s - filter(u) filter(ds)."""
    nx, ny, nz = 24, 24, 8
    x, y = G["n24_x"], G["n24_y"]
    z = np.arange(nz) / nz
    g = [O.FdmPlan(x, True, True), O.FdmPlan(y, False, False), O.FdmPlan(z, True, True)]
    fx, _ = filter_of("n24_t9_p1_b00")
    fy, _ = filter_of("n24_t1_p0_b11")
    rng = np.random.default_rng(3)
    s, v = rng.uniform(-1, 1, nx * ny * nz), rng.uniform(-1, 1, nx * ny * nz)
    for d, f in ((1, fx), (2, fy)):
        got = O.opr_burgers(d, nx, ny, nz, 0, g[d - 1], 0.01, s, v, dealiasing=f)[0]
        d1 = O.opr_partial(d, 1, nx, ny, nz, 0, g[d - 1], s)[0]
        d2 = O.opr_partial(d, 2, nx, ny, nz, 0, g[d - 1], s)[0]
        lines = lambda a: O._to_lines(a, nx, ny, nz, d)      # noqa: E731
        want = 0.01 * d2 - O._from_lines(F.opr_filter_1d(f, lines(v)) * F.opr_filter_1d(f, lines(d1)), nx, ny, nz, d)
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
        assert np.abs(got - O.opr_burgers(d, nx, ny, nz, 0, g[d - 1], 0.01, s, v)[0]).max() > 1e-3
