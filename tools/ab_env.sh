#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of ONE tree under two values of an environment switch (e.g. SW_WG_FOLD 1 0):
# headline / all-resident / sustained steps/s of both, interleaved three times, then one replayed step's kernel timeline
# of each (rocprofv3 --kernel-trace).  Output: gpurun_out/<tag>/ab.txt, kstep_<value>.txt
# usage: bash tools/ab_env.sh <tag> <VAR> <value A> <value B> [workload]
TAG=${1:-ab}; VAR=$2; VA=$3; VB=$4; W=${5:-m1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
B="--no-cpu-baseline --no-other-workloads"
[ $W = m1 ] && N="--steps 40 --warmup 8" || N="--workload $W --steps 12 --warmup 4"
for i in 1 2 3; do for V in $VA $VB; do
  env $VAR=$V python bench.py $N $B 2>/dev/null | grep "^{" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-16s value %.1f resident %.1f sustained %.1f (%.4f ms)' % ('$VAR=$V', r['value'], r['config']['inputs_resident']['steps_s'], r['config']['sustained']['steps_s'], r['config']['sustained']['ms_per_step']))"
done; done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for V in $VA $VB; do
  rm -rf /tmp/ks_$V
  env $VAR=$V timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ks_$V -o t -- python $REPO/bench.py $N $B --no-sustained > /tmp/ks_$V.log 2>&1
  python $REPO/tools/rocpd_step.py $(find /tmp/ks_$V -name '*.db' | head -1) > $OUT/kstep_$V.txt 2>/dev/null
  echo "---- $VAR=$V"; cut -c1-110 $OUT/kstep_$V.txt
done
