#!/bin/bash
# Runs ON THE GPU BOX: bench.py of the working tree under a list of environment settings, same box, interleaved twice.
# usage: bash tools/r3_sweep.sh <tag> "VAR=a VAR=b ..."   (each item one setting; "-" = defaults)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
  for S in "$@"; do
    E=""; [ "$S" != "-" ] && E="$S"
    R=$(env $E timeout 300 python bench.py ${BENCH_ARGS:---steps 300 --warmup 30} --no-cpu-baseline --no-other-workloads --no-sustained 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))")
    echo "$S  $R" | tee -a $OUT/sweep_$TAG.log
  done
done
