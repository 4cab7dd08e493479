#!/bin/bash
# What "the slow state of the box" is: N default bench runs back to back on one box, with the device's own temperature / clock /
# power read-outs (rocm-smi) before each.     gpurun --timeout 600 -- "bash tools/thermal_probe.sh r05 10"
cd "$(dirname "$0")/.."
R=${1:-r05}; N=${2:-10}
O=gpurun_out/$R; mkdir -p "$O"
OUT=$O/thermal_state.txt
: > "$OUT"
smi() {
    rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "Temperature|Power|sclk|mclk|fclk|socclk" | sed 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';' | cut -c1-600
}
T0=$(date +%s.%N)
for i in $(seq 1 "$N"); do
    S=$(smi)
    python3 bench.py --steps 15 --warmup 3 --cpu-sample 0 --no-freeslip-leg 2>/dev/null | grep '^{' > "$O/thermal_run.json"
    python3 - "$O/thermal_run.json" "$i" "$T0" "$S" >> "$OUT" <<'PY'
import json, sys, time
r = json.load(open(sys.argv[1]))
ks = {k["kernel"]: k["avg_ms"] for k in r["kernels"]}
names = ["k_xline<BURGERS>", "k_htile<BURGERS>", "k_ptile<BURGERS>", "k_ode_nn", "k_fftz"]
print("run %2s  t=%6.1f s  ms_per_step %.3f  " % (sys.argv[2], time.time() - float(sys.argv[3]), r["ms_per_step"]) +
      "  ".join("%s %.3f" % (n, ks.get(n, float("nan"))) for n in names) + "  copy16 %.0f GB/s" % r.get("copy_ceiling_own", {}).get("copy16_GBps", float("nan")))
print("        before it: " + sys.argv[4])
PY
done
rm -f "$O/thermal_run.json"
cat "$OUT"
