#!/bin/bash
# Same-box A/B of bench.py: working tree vs the tree under ab_old/ (git archive of a reference commit, built).
# Box-to-box clock differences on the GPU pool are larger than most single optimisations; only runs inside
# one gpurun call compare.   usage (on the GPU box): bash tools/ab.sh [bench args...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { timeout 300 python $1/bench.py --steps 300 --warmup 30 --no-cpu-baseline "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %8.1f steps/s   dominant %.4f ms' % ('$1', d['value'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2 3; do one . "$@"; one ab_old "$@"; done
