#!/bin/bash
# tools/set_probe under counter passes, ONE counter per pass (more than a few per block is refused: "exceeds the capabilities of the hardware", and the
# aborted profiler then hangs until killed); one process each: the sets are drawn from the same seed, the ADDRESSES differ from process to process,
# so every pass prints its own rates and slow sets are compared with fast sets inside a pass.  Output: gpurun_out/r06_set_probe/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_set_probe
mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/tools/set_probe
cd /tmp && export TMPDIR=/tmp
$BIN 16 1 > $OUT/plain.txt 2>&1
for C in TCC_HIT TCC_MISS TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCC_EA0_RDREQ_DRAM_CREDIT_STALL \
         TCC_TAG_STALL TCC_EA0_WRREQ_STALL GRBM_UTCL2_BUSY TCP_PENDING_STALL_CYCLES; do
  rm -rf /tmp/sp_raw
  timeout -s KILL 70 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sp_raw -- $BIN 16 1 > $OUT/$C.txt 2> /tmp/sp.err || { echo "$C failed: $(grep -m1 -i "error\|exceeds" /tmp/sp.err)" >> $OUT/failed.txt; continue; }
  f=$(find /tmp/sp_raw -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${C}_counters.csv
done
ls $OUT
