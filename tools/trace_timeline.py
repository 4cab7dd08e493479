#!/usr/bin/env python3
"""Prints the kernel timeline of the last full Poisson solve in a rocprofv3 --kernel-trace CSV (start, end in us, queue, kernel) and the
substep period (distance between consecutive k_fftx_r2c launches)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
r2c = [i for i, r in enumerate(rows) if "k_fftx_r2c" in r["Kernel_Name"]]
if len(r2c) >= 3:
    per = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6 for a, b in zip(r2c[:-1], r2c[1:])]
    print("substep periods (ms):", " ".join("%.3f" % p for p in per))
i = r2c[-2]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[i:]:
    print("%9.1f %9.1f q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id"), r["Kernel_Name"][:60]))
    if "k_fftx_c2r" in r["Kernel_Name"] and int(r["Start_Timestamp"]) > t0:
        break
