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


def test_substep_traffic_divides_by_the_substeps_of_its_own_table():
    """bench.py's `substep_traffic`: the kernel table may come from a pass of 6 substeps while --steps is 20 (round 5 divided by --steps and
    under-reported the substep's HBM bytes 3.3 x); a substep moves at least one launch of its dominant kernel."""
    sys.path.insert(0, ROOT)
    import bench as B
    per = {"_meta": {"x": 1}, "k_a": 15.0e9, "k_b": 2.0e9}
    kernels = [{"kernel": "k_a", "calls": 6}, {"kernel": "k_b", "calls": 30}, {"kernel": "rocfft", "calls": 6}, {"kernel": "k_unknown", "calls": 6}]
    t = B.substep_traffic(kernels, 6, per, 16.0, (512, 512, 512), dominant_traffic=15.0e9)
    fft = 2.0 * 8.0 * 514 * 512 * 512
    assert abs(t["hbm_bytes_per_step"] - (15.0e9 + 5 * 2.0e9 + fft)) < 1.0
    assert t["consistent"] is True and t["kernels_without_counter_data"] == ["k_unknown"] and t["substeps_in_the_kernel_table"] == 6
    wrong = B.substep_traffic(kernels, 20, per, 16.0, (512, 512, 512), dominant_traffic=15.0e9)       # the round-5 mistake is visible in the line
    assert wrong["consistent"] is False
