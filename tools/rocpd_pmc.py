"""Per-kernel mean PMC counter values from a rocprofv3 --pmc rocpd database."""
import sqlite3
import sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
pmc_view = [t for t in tabs if t in ("counters_collection", "pmc_events", "counters")]
cols = {t: [r[1] for r in db.execute("pragma table_info(%s)" % t)] for t in pmc_view}
if "--schema" in sys.argv:
    print(tabs); print(cols); sys.exit()
rows = db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
per = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for k, c, v, d in rows:
    per[k.split("(")[0][:40]][c] += v
    cnt[k.split("(")[0][:40]].add(d)
names = sorted({c for k in per for c in per[k]})
print("%-40s %5s " % ("kernel", "n") + " ".join("%14s" % n[-14:] for n in names))
for k in sorted(per, key=lambda k: -per[k].get(names[0], 0)):
    n = len(cnt[k])
    print("%-40s %5d " % (k, n) + " ".join("%14.0f" % (per[k][c] / n) for c in names))
