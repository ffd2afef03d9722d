#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench
# command, summarised by tools/make_profiles.sh into gpurun_out/prof_<tag>/summ/ (copy those to profiles/);
# the databases are deleted afterwards (gpurun_out/ is capped at 64 MiB), only the m1 trace db is kept.
# usage: bash tools/collect_profiles.sh <tag>
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads"
timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format rocpd -d $OUT -o sq -- $CMD > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d $OUT -o trace_c4 -- python $REPO/bench.py --workload c4 --steps 6 --warmup 3 --no-cpu-baseline --no-other-workloads > $OUT/trace_c4.log 2>&1
cd $REPO && bash tools/make_profiles.sh $OUT $OUT/summ ${2:-r02}
rm -f $OUT/fetch_results.db $OUT/write_results.db $OUT/sq_results.db $OUT/trace_c4_results.db
tail -1 $OUT/trace.log | cut -c1-300
