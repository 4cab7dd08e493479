#!/bin/bash
# One profiling session on the GPU box (from the repo root): everything profiles/<round>/ and profiles/traffic.json are made from, out of the SAME
# binaries in ONE session.
#     gpurun --timeout 1500 -- "bash tools/profile_round.sh r03 $(git rev-parse --short HEAD)"
# then copy gpurun_out/<round>/{*.csv,*.txt,*.json} to profiles/<round>/ and gpurun_out/<round>/traffic.json to profiles/traffic.json.
# rocprofv3 rules of this pool: the program itself after `--` (no env / bash -c hops), --pmc passes separate from --stats, counters one per pass.
set -u
R=${1:-r05}
COMMIT=${2:-unknown}
ROOT=$(pwd)
O=$ROOT/gpurun_out/$R
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="$ROOT/bench.py"
# 1. the default bench line (what the driver runs) and the same command under the kernel trace
python3 "$B" > "$O/bench_default.json" 2> "$O/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/raw_stats" -- python3 "$B" --steps 6 --warmup 2 --cpu-sample 0 --no-freeslip-leg --no-fortran-host > "$O/bench_under_rocprof.json" 2> "$O/rocprof_stats.err"
python3 "$ROOT/tools/pmc_summary.py" stats "$O/raw_stats" > "$O/rocprofv3_kernel_stats_bench_steps6.csv"
# (the trace of the default command also holds the launches of the placement search, the warm-up and the kernel-table pass: the timed region by itself)
python3 "$ROOT/tools/pmc_summary.py" stats_window "$O/raw_stats" "$O/bench_under_rocprof.json" > "$O/rocprofv3_kernel_stats_bench_steps6_timed_region.csv" 2>> "$O/rocprof_stats.err"
# 2. HBM traffic: two separate counter passes, kernel trace only (without the placement search: the bytes of a launch do not depend on where the arrays live)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/raw_fetch" -- python3 "$B" --steps 3 --warmup 1 --cpu-sample 0 --no-freeslip-leg --no-fortran-host --placement-trials 0 > /dev/null 2> "$O/rocprof_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/raw_write" -- python3 "$B" --steps 3 --warmup 1 --cpu-sample 0 --no-freeslip-leg --no-fortran-host --placement-trials 0 > /dev/null 2> "$O/rocprof_write.err"
python3 "$ROOT/tools/pmc_summary.py" pmc "$O/raw_fetch" "$O/raw_write" "$O/traffic.json" "$COMMIT" > "$O/pmc_hbm_traffic_summary.txt"
# 2b. the clock over a long run ("k_xline slows 5 % under sustained load -- clocks, presumably"): one counter pass of 150 substeps
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$O/raw_clock" -- python3 "$B" --steps 150 --warmup 3 --cpu-sample 0 --no-freeslip-leg --no-fortran-host --placement-trials 0 > "$O/bench_under_clock_pass.json" 2> "$O/rocprof_clock.err"
python3 "$ROOT/tools/pmc_summary.py" clock "$O/raw_clock" > "$O/clock_probe.txt" 2>> "$O/rocprof_clock.err"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES --output-format csv -d "$O/raw_clock2" -- python3 "$B" --steps 150 --warmup 3 --cpu-sample 0 --no-freeslip-leg --no-fortran-host --placement-trials 0 > /dev/null 2>> "$O/rocprof_clock.err"
python3 "$ROOT/tools/pmc_summary.py" clock "$O/raw_clock2" SQ_BUSY_CYCLES >> "$O/clock_probe.txt" 2>> "$O/rocprof_clock.err"
# (third argument "quick": the bench line, its kernel trace and the traffic file only -- after a change that leaves the other lines as they are)
if [ "${3:-}" = "quick" ]; then rm -rf "$O"/raw_*; ls -la "$O"; exit 0; fi
# 3. the north-star's own kernels stand-alone: OPR_Partial_{X,Y,Z}(OPR_P1) at 512^3, kernel trace + the same two counter passes
OPS="$ROOT/tools/bench_ops.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/raw_ops_stats" -- python3 "$OPS" --types P1 --iters 20 > "$O/ops_p1_under_rocprof.txt" 2> "$O/rocprof_ops.err"
python3 "$ROOT/tools/pmc_summary.py" stats "$O/raw_ops_stats" > "$O/rocprofv3_kernel_stats_opr_partial_p1.csv"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/raw_ops_fetch" -- python3 "$OPS" --types P1 --iters 5 > /dev/null 2>> "$O/rocprof_ops.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/raw_ops_write" -- python3 "$OPS" --types P1 --iters 5 > /dev/null 2>> "$O/rocprof_ops.err"
python3 "$ROOT/tools/pmc_summary.py" pmc "$O/raw_ops_fetch" "$O/raw_ops_write" > "$O/pmc_hbm_traffic_opr_partial_p1.txt"
python3 "$OPS" --types P1,BURGERS --iters 20 > "$O/ops_standalone.txt" 2>&1
# 4. the other lines profiles/README.md quotes, same binaries, same session (outside the profiler)
Q="--cpu-sample 0 --no-freeslip-leg --no-fortran-host"
python3 "$B" --loopback 8 --steps 20 --warmup 5 $Q > "$O/bench_loopback8_native.json" 2> /dev/null
TLAB_SLAB_FUSED_X=0 python3 "$B" --loopback 8 --steps 20 --warmup 5 $Q > "$O/bench_loopback8_native_unfused.json" 2> /dev/null
python3 "$B" --loopback 8 --walls freeslip --steps 20 --warmup 5 $Q > "$O/bench_loopback8_native_freeslip.json" 2> /dev/null
TLAB_NEUMANN_PLANES=0 python3 "$B" --loopback 8 --walls freeslip --steps 20 --warmup 5 $Q > "$O/bench_loopback8_native_freeslip_derivative_pass.json" 2> /dev/null
python3 "$B" --decomp 2x4 --slab-driver native --steps 6 --warmup 2 $Q > "$O/bench_decomp2x4_native.json" 2> /dev/null
python3 "$B" --decomp 2x4 --slab-driver python --steps 6 --warmup 2 $Q > "$O/bench_decomp2x4_python.json" 2> /dev/null
"$ROOT/tools/yardstick" 512 > "$O/yardstick.jsonl" 2> /dev/null
# the configuration north_star names: the Fortran mini-driver at the benchmark's size, three processes per route, final fields compared (round 6)
python3 "$ROOT/tools/fortran_host.py" --compare --repeat 3 > "$O/fortran_host.json" 2> "$O/fortran_host.err"
# launches of one slab's size against the single domain's, warm and cold (round 6: what the decomposition overhead is NOT)
python3 "$ROOT/tools/small_launch_probe.py" 2> /dev/null | grep -v amdgpu.ids > "$O/small_launches.txt"
AB=${4:-noab}      # fourth argument "ab": also the A/B lines of earlier rounds' switches
if [ "$AB" = "ab" ]; then
TLAB_HTILE_PERSIST=0 python3 "$B" --steps 20 --warmup 5 $Q > "$O/bench_no_ptile.json" 2> /dev/null
TLAB_HTILE_PERSIST=0 TLAB_HTILE_UNI=0 python3 "$B" --steps 20 --warmup 5 $Q > "$O/bench_no_ptile_no_dual_solve.json" 2> /dev/null
fi
python3 "$B" --walls freeslip --steps 20 --warmup 5 $Q > "$O/bench_freeslip.json" 2> /dev/null
TLAB_NEUMANN_PLANES=0 python3 "$B" --walls freeslip --steps 20 --warmup 5 $Q > "$O/bench_freeslip_derivative_pass.json" 2> /dev/null
python3 "$B" --grid 2048 1024 256 --nscal 3 --steps 6 --warmup 2 $Q > "$O/bench_configs4_rank_share.json" 2> /dev/null
python3 "$B" --grid 1024 512 1024 --steps 6 --warmup 2 $Q > "$O/bench_configs3_one_device.json" 2> /dev/null
if [ "$AB" = "ab" ]; then
# the same two with round 4's routes for long lines switched off (line barriers, float-difference tables, own c2r of p, k_htile<P1> epilogues, k_ptile at 1024)
OFF="TLAB_XLINE_LINE_BARRIERS=0 TLAB_XLINE_FF=0 TLAB_FFTX_C2R_OWN=0 TLAB_P1_HTILE=0 TLAB_PTILE_1024=0"
env $OFF python3 "$B" --grid 2048 1024 256 --nscal 3 --steps 6 --warmup 2 $Q > "$O/bench_configs4_rank_share_long_line_routes_off.json" 2> /dev/null
env $OFF python3 "$B" --grid 1024 512 1024 --steps 6 --warmup 2 $Q > "$O/bench_configs3_one_device_long_line_routes_off.json" 2> /dev/null
TLAB_XLINE_LINE_BARRIERS=0 python3 "$ROOT/tools/bench_xlines.py" 2> /dev/null | grep grid > "$O/xlines_workgroup_barriers.jsonl"
fi
TLAB_PROFILE_REPORT=1 python3 "$ROOT/tools/bench_poisson.py" > "$O/poisson_standalone.txt" 2>&1
if [ "$AB" = "ab" ]; then
TLAB_XLINE_OCC=1 python3 "$B" --steps 20 --warmup 5 $Q > "$O/bench_xline_one_wave_per_simd.json" 2> /dev/null
TLAB_PENCIL_OVERLAP=0 python3 "$B" --decomp 2x4 --slab-driver native --steps 6 --warmup 2 $Q > "$O/bench_decomp2x4_native_literal_sequence.json" 2> /dev/null
"$ROOT/tools/mall_probe" > "$O/mall_probe.jsonl" 2> /dev/null
# which allocations the arrays live on: the placement search off / on, interleaved; the probes behind it
for i in 1 2 3; do
    python3 "$B" --steps 15 --warmup 3 $Q --placement-trials 0 > "$O/bench_placement_off_$i.json" 2> /dev/null
    python3 "$B" --steps 15 --warmup 3 $Q > "$O/bench_placement_on_$i.json" 2> /dev/null
done
python3 "$B" --steps 15 --warmup 3 $Q --time-every-launch > "$O/bench_time_every_launch.json" 2> /dev/null
timeout 300 "$ROOT/tools/placement_probe" > "$O/placement_probe.txt" 2>&1
timeout 300 "$ROOT/tools/placement_probe" 100 2>&1 | grep -v '^  array' > "$O/placement_survey.txt"
fi
# all eight slab ranks' work on one GPU, kernel by kernel
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/raw_lb" -- python3 "$B" --loopback 8 --steps 6 --warmup 2 $Q > "$O/bench_loopback8_under_rocprof.json" 2> "$O/rocprof_lb.err"
python3 "$ROOT/tools/pmc_summary.py" stats "$O/raw_lb" > "$O/rocprofv3_kernel_stats_loopback8_steps6.csv"
python3 "$ROOT/tools/pmc_summary.py" gaps "$O/raw_lb" "$O/bench_loopback8_under_rocprof.json" > "$O/loopback8_timeline.txt" 2>> "$O/rocprof_lb.err"
python3 "$ROOT/tools/bench_xlines.py" 2> /dev/null | grep grid > "$O/xlines.jsonl"
python3 "$ROOT/tools/bench_xlines.py" --exact-uniform --grids 2048x1024x64 2> /dev/null | grep grid > "$O/xlines_equal_rows_2048.jsonl"
# the raw rocprofv3 trees are large: only the summaries travel back
rm -rf "$O"/raw_*
ls -la "$O"
