#!/bin/bash
# temporary: A/B of two builds of the library on one box:  tools/ab_run.sh <out dir> <command...>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; shift
mkdir -p $O
cp tlab_amd/libtlab_amd.so /tmp/new.so
for r in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp tlab_amd/libtlab_amd_old.so tlab_amd/libtlab_amd.so; else cp /tmp/new.so tlab_amd/libtlab_amd.so; fi
    "$@" 2>/dev/null | grep -E "grid|metric" > $O/${v}_$r.jsonl
  done
done
cp /tmp/new.so tlab_amd/libtlab_amd.so
