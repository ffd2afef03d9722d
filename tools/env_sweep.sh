#!/bin/bash
# Runs ON THE GPU BOX: the bench's sustained ms per step under several values of one environment switch, two rounds.
# usage: bash tools/env_sweep.sh <tag> <VAR> "<v1 v2 ...>" [workload]
TAG=${1:-sweep}; VAR=$2; VALS=$3; W=${4:-m1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
B="--no-cpu-baseline --no-other-workloads"
[ $W = m1 ] && N="--steps 40 --warmup 8" || N="--workload $W --steps 12 --warmup 4"
for i in 1 2; do for V in $VALS; do
  env $VAR=$V python bench.py $N $B 2>/dev/null | grep "^{" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); ks={k['name']:k for k in r['roofline']['kernels']}
wp=ks.get('wgrad_partial_kernel',{}).get('us_per_step'); wr=ks.get('wgrad_reduce_kernel',{}).get('us_per_step')
print('%-16s value %.1f resident %.1f sustained %.1f (%.4f ms) eager wgrad_partial %s wgrad_reduce %s' % ('$VAR=$V', r['value'], r['config']['inputs_resident']['steps_s'], r['config']['sustained']['steps_s'], r['config']['sustained']['ms_per_step'], wp, wr))"
done; done | tee $OUT/sweep.txt
