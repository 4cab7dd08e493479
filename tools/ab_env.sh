#!/bin/bash
# A/B of environment switches on one box, interleaved:  tools/ab_env.sh <out name> <VAR> <v1> <v2> [...]  -- runs bench.py twice per value, round-robin,
# and prints ms_per_step + the average launch time of the kernels named in $KERNELS (default: the Burgers kernels) per run.
cd "$(dirname "$0")/.."
O=gpurun_out/$1; VAR=$2; shift 2
mkdir -p "$O"
K=${KERNELS:-"k_xline<BURGERS> k_htile<BURGERS> k_ptile<BURGERS> k_ode_nn"}
for r in 1 2; do
  for v in "$@"; do
    env "$VAR=$v" python bench.py --steps ${STEPS:-15} --warmup 3 --cpu-sample 0 --no-freeslip-leg 2>/dev/null | grep '^{' > "$O/${VAR}_${v}_$r.json"
    python - "$O/${VAR}_${v}_$r.json" "$VAR=$v run $r" $K <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
ks = {k["kernel"]: k["avg_ms"] for k in r["kernels"]}
print("%-28s ms_per_step %.3f  " % (sys.argv[2], r["ms_per_step"]) + "  ".join("%s %.3f" % (n, ks.get(n, float("nan"))) for n in sys.argv[3:]), flush=True)
PY
  done
done
