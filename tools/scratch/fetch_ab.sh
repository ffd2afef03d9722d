#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of the m1 step for the working tree and ab_old/ (per-step HBM bytes by kernel).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="--steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-sustained ${BENCH_ARGS}"
for T in . ab_old; do
  N=$(echo $T | tr -d './'); N=${N:-new}
  rm -rf /tmp/f_$N
  timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/f_$N -o fetch -- python $REPO/$T/bench.py $B > /tmp/f_$N.log 2>&1
  timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/f_$N -o write -- python $REPO/$T/bench.py $B > /tmp/f_$N.w.log 2>&1
  echo "== $N"
  (cd $REPO && python tools/pmc_traffic.py /tmp/f_$N/fetch_results.db /tmp/f_$N/write_results.db /tmp/f_$N.json | head -30)
done
