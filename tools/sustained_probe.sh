#!/bin/bash
# Is "the slow state of the box" a matter of TIME under load or of the PROCESS (its allocations)?  Short runs in separate processes, one long run in one
# process (same total time as many short ones), short runs again; each line: seconds since the start, ms per substep, the bandwidth-bound kernels.
#     gpurun --timeout 600 -- "bash tools/sustained_probe.sh r05"
cd "$(dirname "$0")/.."
R=${1:-r05}
O=gpurun_out/$R; mkdir -p "$O"
OUT=$O/sustained_probe.txt
: > "$OUT"
T0=$(date +%s.%N)
one() {
    python3 bench.py --steps "$1" --warmup 3 --cpu-sample 0 --no-freeslip-leg 2>/dev/null | grep '^{' > "$O/sp_run.json"
    python3 - "$O/sp_run.json" "$2" "$T0" >> "$OUT" <<'PY'
import json, sys, time
r = json.load(open(sys.argv[1]))
ks = {k["kernel"]: k["avg_ms"] for k in r["kernels"]}
names = ["k_xline<BURGERS>", "k_htile<BURGERS>", "k_ptile<BURGERS>", "k_ode_nn"]
print("%-34s t=%6.1f s  steps %5d  ms_per_step %.3f  " % (sys.argv[2], time.time() - float(sys.argv[3]), r["steps"], r["ms_per_step"]) +
      "  ".join("%s %.3f" % (n, ks.get(n, float("nan"))) for n in names))
PY
}
for i in 1 2 3; do one 15 "short run $i (own process)"; done
one 1500 "long run (one process)"
for i in 4 5 6; do one 15 "short run $i (own process)"; done
rm -f "$O/sp_run.json"
rocm-smi --showtemp --showmeminfo vram 2>/dev/null | grep -E "Temperature|VRAM" | sed 's/^GPU\[0\]\s*:\s*//' >> "$OUT"
cat "$OUT"
