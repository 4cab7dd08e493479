cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lb8_raw /tmp/probe_raw
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lb8_raw -- python3 $R/bench.py --loopback 8 --steps 6 --warmup 2 --cpu-sample 0 > /tmp/lb8.out 2>&1; tail -2 /tmp/lb8.out | cut -c1-300
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/probe_raw -- python3 $R/tools/small_launch_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag, d in (("loopback8", "/tmp/lb8_raw"), ("probe", "/tmp/probe_raw")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rd = csv.DictReader(open(f))
        print(tag, "columns:", rd.fieldnames)
        for r in rd:
            if "k_htile<32, 4" in r["Kernel_Name"] or "k_xline<8, 4" in r["Kernel_Name"]:
                key = (r["Kernel_Name"][:70],) + tuple(r.get(c) for c in ("Grid_Size", "Grid_Size_X", "Workgroup_Size", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Scratch_Size"))
                acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for k, v in acc.items():
        v.sort()
        print(tag, k, "n=%d median %.1f us min %.1f max %.1f" % (len(v), v[len(v) // 2], v[0], v[-1]))
PY
