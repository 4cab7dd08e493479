"""tools/pmc_summary.py: the timed region of a bench.py run is cut out of a rocprofv3 kernel trace by LAUNCH COUNT (placement search + warm-up launches
of the dominant kernel come first) -- held here to a synthetic trace with a known answer."""
import csv
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _trace(tmp_path, trials, warmup, steps, extra):
    """substeps of 100 us: k_a 40 us (the dominant kernel), k_b 30 us, idle 30 us; the timed ones run k_a in 50 us instead"""
    rows, t = [], 1000.0
    total = 4 * trials + warmup + steps + extra
    first_timed = 4 * trials + warmup
    for i in range(total):
        timed = first_timed <= i < first_timed + steps
        da = 50e3 if timed else 40e3
        rows.append((t, t + da, "void tlab::k_xline<8, 4, 1, 1, 1, 256, true, true, 2>(tlab::XLineArgs)"))
        rows.append((t + da, t + da + 30e3, "void tlab::k_fftz<1, 8>(tlab::FftzArgs)"))
        t += 100e3
    d = tmp_path / "raw" / "host"
    d.mkdir(parents=True)
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for a, e, n in rows:
            w.writerow(["KERNEL_DISPATCH", n, int(a), int(e)])
    bench = {"steps": steps, "warmup": warmup, "ms_per_step": 0.1, "roofline": {"kernel": "k_xline<BURGERS>", "launches": steps},
             "placement": {"trials": trials} if trials else None}
    (tmp_path / "bench.json").write_text("noise\n" + json.dumps(bench) + "\n")
    return str(tmp_path / "raw"), str(tmp_path / "bench.json")


def test_timed_region_is_found_by_launch_count(tmp_path):
    import pmc_summary as P
    raw, bench = _trace(tmp_path, trials=3, warmup=2, steps=5, extra=7)
    rows, t0, t1, b = P.timed_window(raw, bench)
    assert (t1 - t0) == 5 * 100e3 and t0 == 1000.0 + (4 * 3 + 2) * 100e3
    out = io.StringIO()
    with redirect_stdout(out):
        P.stats_window(raw, bench)
    table = list(csv.reader(io.StringIO(out.getvalue())))
    body = {r[0]: r for r in table[2:]}
    xa = body["void tlab::k_xline<8, 4, 1, 1, 1, 256, true, true, 2>(tlab::XLineArgs)"]
    assert int(xa[1]) == 5 and float(xa[3]) == 50e3            # only the timed launches: 50 us each, not the 40 us of the search / warm-up / table pass
    assert int(body["void tlab::k_fftz<1, 8>(tlab::FftzArgs)"][1]) == 5
    out = io.StringIO()
    with redirect_stdout(out):
        P.gaps(raw, bench)
    text = out.getvalue()
    assert "sum of kernel durations 0.080 ms" in text and "no kernel running 0.020 ms" in text


def test_tags_follow_the_kernel_templates():
    import pmc_summary as P
    assert P.tag("void tlab::k_zslab<16, 4, 2, true>(tlab::ZSlabArgs)") == "k_zslab<BURGERS,B>"
    assert P.tag("void tlab::k_zslab<16, 1, 1>(tlab::ZSlabArgs)") == "k_zslab<P1,A>"
    assert P.tag("void tlab::k_xline<8, 4, 1, 1, 1, 256, true, true, 2>(tlab::XLineArgs)") == "k_xline<BURGERS>"
    assert P.tag("void tlab::k_ptile<32, 32, 16, true, false>(tlab::RTileArgs, long long)") == "k_ptile<BURGERS+div>"
