#!/bin/bash
# Runs ON THE GPU BOX: GPU tests of the working tree, then a same-box A/B of bench.py (working tree vs ab_old/) and
# per-kernel stats of both under rocprofv3.   usage: bash tools/r3_ab.sh <tag> [quick]
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests_$TAG.log
fi
bash tools/ab.sh --no-other-workloads 2>&1 | tee $OUT/ab_$TAG.log
cd /tmp && export TMPDIR=/tmp
for T in . ab_old; do
  N=$(echo $T | tr -d './'); N=${N:-new}
  rm -rf /tmp/ks_$N
  timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/ks_$N -o t -- python $REPO/$T/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-workloads > /tmp/ks_$N.log 2>&1
  python $REPO/tools/rocpd_step.py $(find /tmp/ks_$N -name '*.db' | head -1) > $OUT/kstep_${TAG}_$N.txt 2>/dev/null
  cat $OUT/kstep_${TAG}_$N.txt | cut -c1-110
done
