#!/bin/bash
# One counter per pass (the pool's rule; more per block may be refused) over one OPR_Poisson (tools/bench_poisson.py); prints per kernel the mean per launch.
#     bash tools/pmc_any.sh <out file> COUNTER [COUNTER ...]
set -u
ROOT=$(pwd)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
export TMPDIR=/tmp
cd /tmp
: > "$ROOT/$OUT"
for C in "$@"; do
  rm -rf /tmp/pmc_any_raw
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_any_raw -- python3 "$ROOT/tools/bench_poisson.py" --iters 3 > /dev/null 2> /tmp/pmc_any.err || { echo "$C: rocprofv3 failed: $(grep -m1 -i 'exceeds\|error' /tmp/pmc_any.err)" >> "$ROOT/$OUT"; continue; }
  python3 - "$C" >> "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_any_raw/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == c:
            acc[r['Kernel_Name'][:48]].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if any(s in k for s in ('k_ode_nn', 'k_fftz', 'k_fftx')):
        print("%-34s %-50s launches %3d  mean %.4e" % (c, k, len(v), sum(v) / len(v)))
PY
done
cat "$ROOT/$OUT"
