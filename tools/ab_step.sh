#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of the working tree against ab_old/ (a built `git archive` of the reference commit):
# sustained steps/s of both, interleaved, then one replayed step's kernel timeline of each (rocprofv3).
# usage: bash tools/ab_step.sh <tag> [workload]
TAG=${1:-ab}
W=${2:-m1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
B="--no-cpu-baseline --no-other-workloads"
[ $W = m1 ] && N="--steps 40 --warmup 8" || N="--workload $W --steps 12 --warmup 4"
for i in 1 2 3; do for T in . ab_old; do
  python $T/bench.py $N $B 2>/dev/null | grep "^{" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-7s value %.1f resident %.1f sustained %.1f' % ('$T', r['value'], r['config']['inputs_resident']['steps_s'], r['config']['sustained']['steps_s']))"
done; done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for T in . ab_old; do
  NAME=$(echo $T | tr -d './'); NAME=${NAME:-new}
  rm -rf /tmp/ks_$NAME
  timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ks_$NAME -o t -- python $REPO/$T/bench.py $N $B --no-sustained > /tmp/ks_$NAME.log 2>&1
  python $REPO/tools/rocpd_step.py $(find /tmp/ks_$NAME -name '*.db' | head -1) > $OUT/kstep_$NAME.txt 2>/dev/null
  cut -c1-100 $OUT/kstep_$NAME.txt
done
