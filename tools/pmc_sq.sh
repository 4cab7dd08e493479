#!/bin/bash
# SQ counters of the kernels of one OPR_Poisson (tools/bench_poisson.py), one counter per pass as the pool's rules ask; from the repo root on a GPU box:
#     bash tools/pmc_sq.sh <out file>
# prints, per kernel, the mean of every counter over its launches
set -u
ROOT=$(pwd)
OUT=${1:-gpurun_out/pmc_sq.txt}
mkdir -p "$(dirname "$OUT")"
export TMPDIR=/tmp
cd /tmp
: > "$ROOT/$OUT"
for C in SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT; do
  rm -rf /tmp/pmc_sq_raw
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_sq_raw -- python3 "$ROOT/tools/bench_poisson.py" --iters 3 > /dev/null 2> /tmp/pmc_sq.err || { echo "$C: rocprofv3 failed: $(tail -1 /tmp/pmc_sq.err)" >> "$ROOT/$OUT"; continue; }
  python3 - "$C" >> "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_sq_raw/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == c:
            acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if 'k_ode_nn' in k or 'k_fftz' in k or 'k_fftx' in k:
        print("%-24s %-62s launches %3d  mean %.4e" % (c, k, len(v), sum(v) / len(v)))
PY
done
cat "$ROOT/$OUT"
