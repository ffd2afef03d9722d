#!/bin/bash
# Runs ON THE GPU BOX: one replayed step's kernel timeline under rocprofv3 for each library variant given
# (variants/lib_<name>.so, built by tools/build_variant.sh; "-" = the in-tree library).   usage: bash tools/r3_variants.sh <tag> <name> ...
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  E=""; [ "$V" != "-" ] && E="SW_LIB_PATH=$REPO/variants/lib_$V.so"
  rm -rf /tmp/kv_$V
  env $E timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kv_$V -o t -- python $REPO/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-workloads --no-sustained ${BENCH_ARGS} > /tmp/kv_$V.log 2>&1
  echo "== variant $V" | tee -a $OUT/variants_$TAG.txt
  python $REPO/tools/rocpd_step.py $(find /tmp/kv_$V -name '*.db' | head -1) 2>/dev/null | cut -c1-80 | tee -a $OUT/variants_$TAG.txt
done
