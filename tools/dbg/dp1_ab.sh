# Runs ON THE GPU BOX: the 1-rank data-parallel structure legs (dp1_rccl / dp1_direct) with the working tree's library against
# variants/lib_oldcomm.so (the same objects with an older csrc/sw_comm.hip), interleaved three times.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for LIB in new old; do
  if [ $LIB = old ]; then export SW_LIB_PATH=variants/lib_oldcomm.so SW_ALLREDUCE_CHECK=0; else unset SW_LIB_PATH SW_ALLREDUCE_CHECK; fi
  python bench.py --side-legs dp1 --side-out /tmp/dp1_$LIB.json --ref-ms 0.3700 2>/dev/null
  python -c "
import json; r=json.load(open('/tmp/dp1_$LIB.json')); print('$LIB', 'rccl %.4f ms' % r['dp1_rccl']['ms_per_step'], 'direct %.4f ms' % r['dp1_direct'].get('ms_per_step', -1), r['dp1_direct'].get('error',''))"
done; done
