#!/usr/bin/env python3
"""Reads bench.py's JSON line on stdin and prints ms_per_step and the per-kernel table (calls per step, average ms, ms per step)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    steps = d["steps"]
    print("ms_per_step %.3f  host_issue %.3f" % (d["ms_per_step"], d.get("host_issue_ms_per_step", 0.0)))
    tot = 0.0
    for e in d.get("kernels", []):
        tot += e["total_ms"] / steps
        print("  %-26s %6.1f/step  avg %.4f ms  %.3f ms/step" % (e["kernel"], e["calls"] / steps, e["avg_ms"], e["total_ms"] / steps))
    print("  sum of kernels per step %.3f ms" % tot)
