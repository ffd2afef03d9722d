"""Kernel-time summary (what `rocprofv3 --stats` prints) from a rocprofv3 rocpd sqlite database.
    python tools/rocpd_stats.py gpurun_out/prof_x/m1_results.db [skip_first_n_per_kernel] > profiles/...txt
"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
agg = defaultdict(list)
for name, s, e in rows:
    agg[name].append(e - s)
tot = sum(sum(v[skip:]) for v in agg.values())
print("%-72s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][skip:])):
    v = v[skip:] or [0]
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:72]
    print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3,
                                                         min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / max(tot, 1)))
print("total kernel time %.1f us over %d dispatches; wall span %.1f us" % (tot / 1e3, len(rows),
      (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0))
