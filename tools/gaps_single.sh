cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/g_raw
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/g_raw -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-freeslip-leg --no-fortran-host > /tmp/g_bench.json 2>/dev/null
python3 $R/tools/pmc_summary.py gaps /tmp/g_raw /tmp/g_bench.json | head -70
