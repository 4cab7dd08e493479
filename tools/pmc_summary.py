#!/usr/bin/env python3
"""Turns rocprofv3 outputs into the summaries kept under profiles/:

    python tools/pmc_summary.py stats  <dir of `rocprofv3 --kernel-trace --stats --output-format csv`>  > profiles/rNN/rocprofv3_kernel_stats_*.csv
    python tools/pmc_summary.py pmc    <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [traffic.json to write [commit]]

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE KiB (gfx950: FETCH_SIZE counts half of the bytes of wide coalesced reads,
MI355X_MICROARCH.md section HBM; calibrated on a 4-array streaming kernel, profiles/README.md)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

MODES = {1: "P1", 2: "P2", 3: "P2_P1", 4: "BURGERS", 5: "P2_D1IN", 6: "BURGERS_D1IN"}


def tag(name):
    """rocprof kernel name -> the tag the library's own profiler (and bench.py) uses."""
    name = re.sub(r"^void\s+", "", name).replace("tlab::", "")
    m = re.match(r"k_htile<(\d+), 4, (\d+), (\d+), true\b", name)
    if m:
        return "k_htile<BURGERS+div>"
    m = re.match(r"k_ptile<(\d+), (\d+), (\d+), (true|false)", name)
    if m:
        return "k_ptile<BURGERS+div>" if m.group(4) == "true" else "k_ptile<BURGERS>"
    m = re.match(r"k_(xline|rtile|htile)<(\d+), (\d+)", name)
    if m:
        return "k_%s<%s>" % (m.group(1), MODES.get(int(m.group(3)), m.group(3)))
    m = re.match(r"k_zslab<(\d+), (\d+), (\d+)(?:, (?:true|false))?>", name)
    if m:
        return "k_zslab<%s,%s>" % (MODES.get(int(m.group(2)), m.group(2)), "A" if m.group(3) == "1" else "B")
    m = re.match(r"k_int1<(\d+), (\d+), (\d+)", name)
    if m:
        return "k_int1<%s>" % {"0": "field", "1": "linear", "2": "unit"}[m.group(3)]
    if name.startswith("k_fftx_c2r<true>"):
        return "k_fftx_c2r<final>"
    m = re.match(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:60]


def find(d, pattern):
    r = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not r:
        sys.exit("no %s under %s" % (pattern, d))
    return r


def stats(d):
    rows = []
    for f in find(d, "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r.get("Name"), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")])


def counters(d, cname):
    acc = defaultdict(lambda: [0.0, 0])
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def source_hash():
    """sha256 over the kernel sources the library is built from (tlab_amd/csrc/*.{hip,cpp,hpp}, sorted by name): the stamp bench.py recomputes before it
    quotes profiles/traffic.json -- counter data measured on other sources is nulled, whatever the commit says."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tlab_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.cpp")) + glob.glob(os.path.join(root, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc(dfetch, dwrite, tjson=None, commit=None):
    fe, wr = counters(dfetch, "FETCH_SIZE"), counters(dwrite, "WRITE_SIZE")
    out = {}
    print("%-44s %-22s %8s %20s %20s  HBM bytes/launch = (2*FETCH+WRITE) KiB" % ("kernel (rocprof name)", "bench tag", "launches", "FETCH_SIZE[KiB]/launch", "WRITE_SIZE[KiB]/launch"))
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        if "tlab" not in k and not k.startswith("k_") and "void k_" not in k:
            continue
        f = fe[k][0] / fe[k][1]
        w = wr[k][0] / wr[k][1] if k in wr and wr[k][1] else float("nan")
        b = (2 * f + w) * 1024.0
        t = tag(k)
        short = re.sub(r"^void\s+", "", k).replace("tlab::", "")[:43]
        print("%-44s %-22s %8d %20.1f %20.1f  %.4e" % (short, t, fe[k][1], f, w, b))
        if t not in out or fe[k][1] > out[t][1]:
            out[t] = (b, fe[k][1])
    if tjson:
        import datetime
        doc = {"_meta": {"commit": commit, "source_hash": source_hash(), "session": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "kernels": len(out),
                         "what": "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB from two separate rocprofv3 --pmc passes of `bench.py` on the "
                                 "binaries of that commit (tools/profile_round.sh); regenerated with the rocprofv3 kernel-stats CSV of the same session"}}
        doc.update({k: v[0] for k, v in out.items()})
        json.dump(doc, open(tjson, "w"), indent=1)


def clock(d, cname="GRBM_GUI_ACTIVE", kernels=("k_xline<BURGERS>", "k_htile<BURGERS>", "k_ptile<BURGERS>", "k_ode_nn", "k_fftz")):
    """cycles per nanosecond of every launch of the named kernels over the course of ONE long bench run (one --pmc pass with the kernel trace): does the
    clock fall under sustained load?  Prints, per kernel, launches, duration and counter / duration for the first and the last
    quarter of the run and in between."""
    disp = {}
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            did = r.get("Dispatch_Id") or r.get("Dispatch_ID") or r.get("dispatch_id")
            t0 = float(r.get("Start_Timestamp") or r.get("Start_Time") or 0.0)
            t1 = float(r.get("End_Timestamp") or r.get("End_Time") or 0.0)
            disp[did] = (r.get("Kernel_Name", ""), t0, t1)
    per = defaultdict(list)
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname:
                continue
            did = r.get("Dispatch_Id") or r.get("Dispatch_ID") or r.get("dispatch_id")
            if did not in disp:
                continue
            name, t0, t1 = disp[did]
            t = tag(name)
            if t in kernels and t1 > t0:
                per[t].append((t0, t1 - t0, float(r["Counter_Value"])))
    print("# %s / duration per launch (counter units per ns; the absolute scale depends on how many instances of the counter the profiler sums -- the CHANGE "
          "over the run is what is read); one bench.py run under rocprofv3 --kernel-trace --pmc %s" % (cname, cname))
    print("%-22s %8s | %-34s | %-34s | %-34s" % ("kernel", "launches", "first quarter: ms, counter/ns", "middle half", "last quarter"))
    for t in kernels:
        v = sorted(per.get(t, []))
        if len(v) < 8:
            continue
        q = len(v) // 4
        parts = (v[:q], v[q:len(v) - q], v[len(v) - q:])
        cells = []
        for pt in parts:
            ms = sum(x[1] for x in pt) / len(pt) * 1e-6
            cl = sum(x[2] for x in pt) / sum(x[1] for x in pt)
            cells.append("%.4f ms  %10.4f" % (ms, cl))
        print("%-22s %8d | %-34s | %-34s | %-34s" % (t, len(v), cells[0], cells[1], cells[2]))


def timed_window(d, bench_json):
    """(rows, t0, t1, bench line): the kernel trace of one bench.py run and the bounds of its TIMED REGION.  The trace of the default command also holds the
    launches of the placement search (tlab_dns_place_arrays: 4 substeps per trial on assignments the run does not use), the warm-up, the empty-queue
    substep and the pass that fills the kernel table; the timed region is found by COUNT: the dominant kernel (roofline.kernel of the bench line of
    the same run, `lpp` launches per substep) is launched 4 lpp times per trial, then lpp warmup times, then lpp steps times -- the window runs from
    the start of the first of those to the start of the launch after the last."""
    b = json.loads([ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")][-1])
    steps, warmup = int(b["steps"]), int(b["warmup"])
    dom = b["roofline"]["kernel"]
    lpp = max(1, int(b["roofline"]["launches"]) // steps)
    trials = int((b.get("placement") or {}).get("trials", 0))
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            rows.append((float(r.get("Start_Timestamp") or 0.0), float(r.get("End_Timestamp") or 0.0), r.get("Kernel_Name", "")))
    rows.sort()
    doms = [r for r in rows if tag(r[2]) == dom]
    i0 = lpp * (4 * trials + warmup)
    if len(doms) <= i0 + lpp * steps:
        sys.exit("timed_window: %d launches of %s in the trace, %d expected before the end of the timed region" % (len(doms), dom, i0 + lpp * steps + 1))
    return rows, doms[i0][0], doms[i0 + lpp * steps][0], b


def stats_window(d, bench_json):
    """The kernel statistics of the timed region of one bench.py run under --kernel-trace (same columns as `stats`)."""
    rows, t0, t1, b = timed_window(d, bench_json)
    steps = int(b["steps"])
    acc = defaultdict(list)
    for a, e, name in rows:
        if a >= t0 and a < t1:
            acc[name].append(e - a)
    tot = sum(sum(v) for v in acc.values())
    w = csv.writer(sys.stdout)
    w.writerow(["# launches that START inside the timed region of the run (%d substeps, window %.3f ms = %.3f ms per substep; the bench line of the same run says %.3f)"
                % (steps, (t1 - t0) * 1e-6, (t1 - t0) * 1e-6 / steps, b["ms_per_step"])])
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), int(sum(v)), "%.1f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / tot), int(min(v)), int(max(v))])


def gaps(d, bench_json):
    """per-tag kernel time per substep and the idle time of the device inside the timed region (timed_window) of one bench.py run under --kernel-trace,
    and its last substep launch by launch."""
    rows, t0, t1, b = timed_window(d, bench_json)
    steps, ms = int(b["steps"]), float(b["ms_per_step"])
    win = [r for r in rows if r[0] >= t0 and r[0] < t1]
    per = defaultdict(lambda: [0.0, 0])
    for a, e, name in win:
        acc = per[tag(name)]
        acc[0] += e - a
        acc[1] += 1
    busy, cur0, cur1 = 0.0, None, None
    for a, e, _ in win:
        if cur1 is None or a > cur1:
            if cur1 is not None:
                busy += cur1 - cur0
            cur0, cur1 = a, e
        else:
            cur1 = max(cur1, e)
    busy += (cur1 - cur0) if cur1 is not None else 0.0
    tot = sum(v[0] for v in per.values())
    span = (t1 - t0) * 1e-6 / steps
    print("# timed region: %d substeps, %.3f ms each in the trace (the bench line of the same run: %.3f); kernels in it: %d" % (steps, span, ms, len(win)))
    print("# per substep: sum of kernel durations %.3f ms, device busy (union of the kernels' intervals) %.3f ms, no kernel running %.3f ms" %
          (tot * 1e-6 / steps, busy * 1e-6 / steps, span - busy * 1e-6 / steps))
    print("%-28s %10s %12s %12s" % ("kernel", "launches", "ms per step", "us / launch"))
    for t, v in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("%-28s %10.1f %12.4f %12.2f" % (t, v[1] / steps, v[0] * 1e-6 / steps, v[0] * 1e-3 / v[1]))
    print("# the last substep of the timed region, launch by launch: start [us from the first], duration [us], idle before it [us] (negative: it overlaps an "
          "earlier kernel), kernel")
    last = [r for r in win if r[0] >= t1 - (t1 - t0) / steps]
    if last:
        z, hi = last[0][0], last[0][0]
        for a, e, name in last:
            print("%10.1f %9.1f %8.1f  %s" % ((a - z) * 1e-3, (e - a) * 1e-3, (a - hi) * 1e-3, tag(name)))
            hi = max(hi, e)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "stats_window":
        stats_window(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 4 and sys.argv[1] == "gaps":
        gaps(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 3 and sys.argv[1] == "clock":
        clock(sys.argv[2], *(sys.argv[3:4]))
    elif len(sys.argv) >= 3 and sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None, sys.argv[5] if len(sys.argv) > 5 else None)
    else:
        sys.exit(__doc__)
