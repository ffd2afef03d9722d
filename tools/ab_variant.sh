#!/bin/bash
# Runs ON THE GPU BOX: the default library against a variant built by tools/build_variant.sh (variants/lib_<name>.so,
# selected with SW_LIB_PATH), same box, interleaved: steps/s, then one replayed step's kernel timeline of each.
# usage: bash tools/ab_variant.sh <variant name> [workload]
V=$1
W=${2:-m1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_$V
mkdir -p $OUT
cd $REPO
B="--no-cpu-baseline --no-other-workloads"
[ $W = m1 ] && N="--steps 40 --warmup 8" || N="--workload $W --steps 12 --warmup 4"
for i in 1 2 3; do for L in default $V; do
  [ $L = default ] && unset SW_LIB_PATH || export SW_LIB_PATH=$REPO/variants/lib_$L.so
  python bench.py $N $B 2>/dev/null | grep "^{" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-8s value %.1f resident %.1f sustained %.1f' % ('$L', r['value'], r['config']['inputs_resident']['steps_s'], r['config']['sustained']['steps_s']))"
done; done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for L in default $V; do
  [ $L = default ] && unset SW_LIB_PATH || export SW_LIB_PATH=$REPO/variants/lib_$L.so
  rm -rf /tmp/ks_$L
  timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ks_$L -o t -- python $REPO/bench.py $N $B --no-sustained > /tmp/ks_$L.log 2>&1
  python $REPO/tools/rocpd_step.py $(find /tmp/ks_$L -name '*.db' | head -1) > $OUT/kstep_$L.txt 2>/dev/null
  echo "== $L"; cut -c1-100 $OUT/kstep_$L.txt | tail -18
done
