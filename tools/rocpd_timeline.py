"""Timeline of the last N kernel dispatches of a rocprofv3 rocpd database (start offset, duration, grid)."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
prev_end = rows[0][1]
for name, s, e, g, w in rows:
    print("%-44s +%9.1f  dur %8.1f us  gap %6.1f  WGs %d" % (name.split("(")[0][:44], (s - t0) / 1e3, (e - s) / 1e3,
                                                         (s - prev_end) / 1e3, g // max(w, 1)))
    prev_end = e
