#!/bin/bash
# ON THE GPU BOX: L2 traffic of the wgrad_partial launches of a training step, per launch kind (grid size).
# usage: bash tools/pmc_wgrad.sh <tag>    (env: anything bench.py / the library reads)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw_$1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/pw_$1 -o t -- python $REPO/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-workloads > /tmp/pw_$1.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
db = sqlite3.connect(glob.glob("/tmp/pw_$1/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
gcol = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = "select kernel_name, counter_name, value, dispatch_id%s from counters_collection" % ((", " + gcol) if gcol else "")
per = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for row in db.execute(q):
    k, c, v, d = row[:4]
    if "wgrad_partial" not in k: continue
    key = row[4] if gcol else 0
    per[key][c] += v; cnt[key].add(d)
for key in sorted(per):
    n = len(cnt[key])
    print("grid %s (%d launches): " % (key, n) + "  ".join("%s %.0f" % (c, per[key][c] / n) for c in sorted(per[key])))
PY
tail -2 /tmp/pw_$1.log | cut -c1-200
