#!/bin/bash
# usage (GPU box): bash tools/sweep.sh ENVVAR "v1 v2 v3" [bench args]   - same-box sweep of a tuning knob
cd ${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=%-8s %8.1f steps/s' % ('$v', d['value']))"
done; done
