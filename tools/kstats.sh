#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations of a short bench run under rocprofv3 for each SW_TILE_MODE given.
# usage: bash tools/kstats.sh <tag> <mode> [<mode> ...]   ->  gpurun_out/kstats_<tag>_m<mode>.txt
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in "$@"; do
  rm -rf /tmp/ks_$M
  SW_TILE_MODE=$M timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ks_$M -o t -- python $REPO/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-workloads ${BENCH_ARGS} > /tmp/ks_$M.log 2>&1
  python $REPO/tools/rocpd_stats.py $(find /tmp/ks_$M -name '*.db' | head -1) > $OUT/kstats_${TAG}_m$M.txt
  python $REPO/tools/rocpd_step.py $(find /tmp/ks_$M -name '*.db' | head -1) > $OUT/kstep_${TAG}_m$M.txt 2>/dev/null
  echo "== mode $M"; head -24 $OUT/kstats_${TAG}_m$M.txt | cut -c1-150
done
