#!/bin/bash
# Post-process <dir>/*.db (rocprofv3 rocpd databases) into text/json summaries in <outdir>, named per round.
# usage: bash tools/make_profiles.sh <dir> <outdir> <round-prefix, e.g. r01>
D=$1; O=$2; R=$3
mkdir -p $O
python tools/rocpd_stats.py $D/trace_results.db > $O/${R}_m1_kernel_stats.txt
python tools/rocpd_step.py $D/trace_results.db > $O/${R}_m1_step_timeline.txt
python tools/rocpd_stats.py $D/trace_c4_results.db > $O/${R}_c4_kernel_stats.txt
python tools/pmc_traffic.py $D/fetch_results.db $D/write_results.db $O/${R}_pmc_m1.json > /dev/null
{ python tools/rocpd_pmc.py $D/fetch_results.db; python tools/rocpd_pmc.py $D/write_results.db; } > $O/${R}_m1_hbm_pmc.txt
[ -f $D/sq_results.db ] && python tools/rocpd_pmc.py $D/sq_results.db > $O/${R}_m1_sq_pmc.txt
ls -la $O
