#!/bin/bash
# Kernel trace of all eight slab ranks' work on one GPU (bench.py --loopback 8) next to the single domain's: where the decomposition overhead sits.
#     gpurun --timeout 900 -- "bash tools/profile_loopback.sh r05"
set -u
R=${1:-r05}
P=${2:-8}
ROOT=$(pwd)
O=$ROOT/gpurun_out/$R
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="$ROOT/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/raw_lb" -- python3 "$B" --loopback "$P" --steps 6 --warmup 2 --cpu-sample 0 --no-freeslip-leg > "$O/bench_loopback${P}_under_rocprof.json" 2> "$O/rocprof_lb.err"
python3 "$ROOT/tools/pmc_summary.py" stats "$O/raw_lb" > "$O/rocprofv3_kernel_stats_loopback${P}_steps6.csv"
python3 "$ROOT/tools/pmc_summary.py" gaps "$O/raw_lb" "$O/bench_loopback${P}_under_rocprof.json" > "$O/loopback${P}_timeline.txt" 2>> "$O/rocprof_lb.err"
rm -rf "$O"/raw_lb
ls -la "$O" | tail -5
