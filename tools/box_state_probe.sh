#!/bin/bash
# The second half of "the slow state of the box": with the placement search on, runs back to back still fall from ~15.9 to ~17.0 ms per substep after a
# while on some boxes (every assignment of the search is slow then).  N runs back to back with rocm-smi sampled DURING each run (clocks, power,
# temperatures under load), a pause, M more runs.      gpurun --timeout 900 -- "bash tools/box_state_probe.sh r05 14 90 4"
cd "$(dirname "$0")/.."
R=${1:-r05}; N=${2:-14}; PAUSE=${3:-90}; M=${4:-4}
O=gpurun_out/$R; mkdir -p "$O"
OUT=$O/box_state_probe.txt
: > "$OUT"
T0=$(date +%s.%N)
sample() {      # while the bench runs: the busiest read-out of a few samples
    for k in 1 2 3 4 5 6; do
        sleep 1.2
        rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "junction|memory\)|Power|fclk|mclk|sclk" | sed 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';' >> "$O/smi_samples.txt"
        echo >> "$O/smi_samples.txt"
    done
}
one() {
    : > "$O/smi_samples.txt"
    sample &
    SP=$!
    python3 bench.py --steps 60 --warmup 3 --cpu-sample 0 --no-freeslip-leg 2>/dev/null | grep '^{' > "$O/bs_run.json"
    wait $SP
    python3 - "$O/bs_run.json" "$1" "$T0" "$O/smi_samples.txt" >> "$OUT" <<'PY'
import json, sys, time, re
r = json.load(open(sys.argv[1]))
ks = {k["kernel"]: k["avg_ms"] for k in r["kernels"]}
p = r.get("placement") or {}
best = ""
pw = -1
for ln in open(sys.argv[4]).read().splitlines():
    m = re.search(r"Power \(W\): ([0-9.]+)", ln)
    if m and float(m.group(1)) > pw:
        pw, best = float(m.group(1)), ln
best = re.sub(r"Temperature \(Sensor (\w+)\) \(C\)", r"T_\1", best).replace("clock level", "").replace("Current Socket Graphics Package Power (W)", "P[W]")
print("%-16s t=%6.1f s  ms_per_step %.3f  search: first %.3f best %.3f worst %.3f  k_xline<BURGERS> %.3f  k_ode_nn %.3f | under load: %s" %
      (sys.argv[2], time.time() - float(sys.argv[3]), r["ms_per_step"], p.get("ms_first", 0), p.get("ms_best", 0), p.get("ms_worst", 0), ks.get("k_xline<BURGERS>", 0),
       ks.get("k_ode_nn", 0), best))
PY
}
for i in $(seq 1 "$N"); do one "run $i"; done
echo "--- $PAUSE s idle ---" >> "$OUT"
sleep "$PAUSE"
for i in $(seq 1 "$M"); do one "after pause $i"; done
rm -f "$O/bs_run.json" "$O/smi_samples.txt"
cat "$OUT"
