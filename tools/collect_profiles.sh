#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel traces + SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters) of
# the default bench command for m1 AND c4, summarised into gpurun_out/prof_<tag>/summ/ (copy those to profiles/);
# the databases are deleted afterwards (gpurun_out/ is capped at 64 MiB).
# usage: SW_COMMIT=<short sha> bash tools/collect_profiles.sh <tag> <round-prefix, e.g. r04>
#   e.g. gpurun -- "SW_COMMIT=$(git rev-parse --short HEAD) bash tools/collect_profiles.sh r04 r04"   (the box has no .git)
TAG=${1:-r03}
R=${2:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
S=$OUT/summ
mkdir -p $S
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-other-workloads --no-sustained"
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"
for W in m1 c4; do
  if [ $W = m1 ]; then N="--steps 30 --warmup 5"; else N="--workload c4 --steps 6 --warmup 3"; fi
  CMD="python $REPO/bench.py $N $B"
  timeout 500 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_$W -o trace -- $CMD > $OUT/trace_$W.log 2>&1
  timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/p_$W -o fetch -- $CMD > $OUT/fetch_$W.log 2>&1
  timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/p_$W -o write -- $CMD > $OUT/write_$W.log 2>&1
  timeout 700 rocprofv3 --kernel-trace --pmc $SQ --output-format rocpd -d /tmp/p_$W -o sq -- $CMD > $OUT/sq_$W.log 2>&1
  cd $REPO
  python tools/rocpd_stats.py /tmp/p_$W/trace_results.db > $S/${R}_${W}_kernel_stats.txt
  python tools/rocpd_step.py /tmp/p_$W/trace_results.db > $S/${R}_${W}_step_timeline.txt
  python tools/pmc_traffic.py /tmp/p_$W/fetch_results.db /tmp/p_$W/write_results.db $S/${R}_pmc_$W.json /tmp/p_$W/sq_results.db > $S/${R}_${W}_step_traffic.txt
  { python tools/rocpd_pmc.py /tmp/p_$W/fetch_results.db; python tools/rocpd_pmc.py /tmp/p_$W/write_results.db; } > $S/${R}_${W}_hbm_pmc.txt
  python tools/rocpd_pmc.py /tmp/p_$W/sq_results.db > $S/${R}_${W}_sq_pmc.txt
  tail -1 $OUT/trace_$W.log | cut -c1-400
  cd /tmp
done
ls -la $S
