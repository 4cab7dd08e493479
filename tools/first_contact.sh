#!/bin/bash
# First contact with a multi-GPU MI355X node: nothing but world-size-1 RCCL has ever executed in this repository, so the
# first run on N > 1 GPUs must be diagnosable step by step.  Each step prints ONE line "FIRST_CONTACT <step> PASS|FAIL <detail>" and writes its
# full output to $OUT/<step>.log; the script stops at nothing (later steps still run) and exits non-zero when any step failed.
#
#   tools/first_contact.sh [N]        N = number of GPUs to use (default: all visible, at most 8)
#
# Steps (each through the product libraries; torch.distributed only carries the ncclUniqueId and the control reductions):
#   0 devices      : rocm-smi / torch see N GPUs
#   1 trp2         : N = 2 transposition round trip through libtlab_amd_comm.so (I and K, real + complex, fp64 and fp32 wire), bit-exact
#                    against the closed form (tools/native_comm_check.py, no torch in the process; identifier through a file)
#   2 trpN         : the same with all N ranks as 1 x N and, when N is even, as 2 x N/2 (x and z communicators by ncclCommSplit)
#   3 slab2, slabN : z-slab substep (native C++ driver, RCCL transport) against the single domain each rank computes redundantly (tools/dist_check.py)
#   4 bench1..N    : bench.py --gpus 1 / 2 / 4 / 8 exactly as the driver launches it; the line's config.rccl_ranks must equal N
#   5 pencil2xK    : (N >= 4, even) bench.py --gpus N --decomp 2x(N/2): the native x/z pencil driver over the world / x / z communicators
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=${FIRST_CONTACT_OUT:-$ROOT/gpurun_out/first_contact}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
N=${1:-$NGPU}
[ "$N" -gt 8 ] && N=8
fail=0
say() { echo "FIRST_CONTACT $1 $2 $3"; [ "$2" = "FAIL" ] && fail=1; return 0; }
port() { python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])"; }

# ---- 0: devices ----
if [ "$NGPU" -ge "$N" ] && [ "$N" -ge 1 ]; then say devices PASS "torch sees $NGPU GPU(s); using $N"; else say devices FAIL "torch sees $NGPU GPU(s), $N asked"; fi
(rocm-smi --showtopo 2>&1 || true) > "$OUT/topology.log"

# ---- 1, 2: transpositions through the native library, one process per GPU, no torch ----
trp() {     # $1 = step name, $2 = ranks, $3 = npro_i
    local name=$1 P=$2 NI=$3 id="$OUT/$1.id" pids=() rc=0
    rm -f "$id" "$id.tmp"
    for r in $(seq 0 $((P - 1))); do
        timeout 300 python3 tools/native_comm_check.py --nranks "$P" --rank "$r" --npro-i "$NI" --idfile "$id" --device "$r" > "$OUT/$name.rank$r.log" 2>&1 &
        pids+=($!)
    done
    for p in "${pids[@]}"; do wait "$p" || rc=1; done
    cat "$OUT/$name".rank*.log > "$OUT/$name.log"
    local okc
    okc=$(grep -c "native comm ok" "$OUT/$name.log" || true)
    if [ "$rc" -eq 0 ] && [ "$okc" -eq "$P" ]; then say "$name" PASS "$P ranks as ${NI}x$((P / NI)): I/K transpositions fp64 + fp32 wire bit-exact, all-reduce"; else say "$name" FAIL "rc=$rc ok=$okc/$P (see $OUT/$name.rank*.log)"; fi
}
if [ "$N" -ge 2 ]; then trp trp2 2 1; else trp trp1 1 1; say trp2 SKIP "needs 2 GPUs (ran the one-rank form instead)"; fi
if [ "$N" -gt 2 ]; then trp "trp${N}_1x$N" "$N" 1; fi
if [ "$N" -ge 4 ] && [ $((N % 2)) -eq 0 ]; then trp "trp${N}_2x$((N / 2))" "$N" 2; fi

# ---- 3: slab substep against the single domain ----
slab() {    # $1 = ranks
    local P=$1 name="slab$1" log="$OUT/slab$1.log"
    TLAB_DIST_BOOTSTRAP=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$P" --master-addr 127.0.0.1 --master-port "$(port)" \
        tools/dist_check.py --driver native --nx 128 --ny 48 --nz $((64 * P)) > "$log" 2>&1
    local rc=$?
    local line
    line=$(grep "DIST_CHECK" "$log" | tail -1)
    if [ "$rc" -eq 0 ] && echo "$line" | grep -q " OK" && echo "$line" | grep -q "backend=nccl"; then say "$name" PASS "$line"; else say "$name" FAIL "rc=$rc ${line:-no DIST_CHECK line} (see $log)"; fi
}
if [ "$N" -ge 2 ]; then slab 2; else slab 1; fi
if [ "$N" -gt 2 ]; then slab "$N"; fi

# ---- 4: the bench contract at 1 / 2 / 4 / 8 ----
for P in 1 2 4 8; do
    [ "$P" -gt "$N" ] && break
    log="$OUT/bench$P.log"
    if [ "$P" -eq 1 ]; then
        timeout 900 python3 bench.py --gpus 1 --steps 12 --warmup 3 --cpu-sample 0 --no-freeslip-leg > "$log" 2>&1
    else
        timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$P" --master-addr 127.0.0.1 --master-port "$(port)" bench.py --gpus "$P" --steps 12 --warmup 3 > "$log" 2>&1
    fi
    rc=$?
    verdict=$(python3 - "$log" "$P" <<'EOF'
import json, sys
log, P = sys.argv[1], int(sys.argv[2])
lines = [l for l in open(log, errors="replace").read().splitlines() if l.startswith("{")]
if len(lines) != 1:
    print("FAIL %d JSON lines" % len(lines)); sys.exit()
r = json.loads(lines[0]); c = r["config"]
ok = r["n_gpus"] == P and c["fields_finite"] and (P == 1 or c.get("rccl_ranks") == P)
print("%s n_gpus=%d rccl_ranks=%s driver=%s ms_per_step=%.3f value=%.4g" % ("PASS" if ok else "FAIL", r["n_gpus"], c.get("rccl_ranks"), c.get("driver"), r["ms_per_step"], r["value"]))
EOF
)
    if [ "$rc" -eq 0 ] && [ "${verdict%% *}" = "PASS" ]; then say "bench$P" PASS "${verdict#* }"; else say "bench$P" FAIL "rc=$rc ${verdict#* } (see $log)"; fi
done
# ---- 5: the x/z pencil decomposition of BASELINE configs[3] on all N ranks (native driver, transpositions started ahead of independent launches) ----
if [ "$N" -ge 4 ] && [ $((N % 2)) -eq 0 ]; then
    log="$OUT/pencil2x$((N / 2)).log"
    timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$(port)" bench.py --gpus "$N" --decomp "2x$((N / 2))" \
        --slab-driver native --steps 6 --warmup 2 > "$log" 2>&1
    rc=$?
    line=$(grep '^{' "$log" | tail -1)
    if [ "$rc" -eq 0 ] && echo "$line" | grep -q '"fields_finite": true' && echo "$line" | grep -q "\"rccl_ranks\": $N"; then
        say "pencil2x$((N / 2))" PASS "$(echo "$line" | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print("ms_per_step=%.3f rccl_ranks=%s" % (r["ms_per_step"], r["config"]["rccl_ranks"]))')"
    else
        say "pencil2x$((N / 2))" FAIL "rc=$rc (see $log)"
    fi
fi
if [ "$fail" -eq 0 ]; then echo "FIRST_CONTACT all PASS"; else echo "FIRST_CONTACT some steps FAILED (logs: $OUT)"; fi
exit $fail
