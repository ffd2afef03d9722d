"""Dispatch timeline of ONE hipGraph-replayed training step (from one stage_step_kernel to the next) out of a
rocprofv3 rocpd database of `bench.py`, plus the step period over the timed region.
    python tools/rocpd_step.py gpurun_out/prof_x/trace_results.db [first kernel of a step] > profiles/rNN_m1_step_timeline.txt"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
first = sys.argv[2] if len(sys.argv) > 2 else "stage_step"      # the kernel a step starts with (wide path: traj4d_kernel)
idx = [i for i, r in enumerate(rows) if r[0].startswith(first)]
# graph-mode steps are back to back (period < 1.2 x median); the eager roofline pass at the end is slower
starts = np.array([rows[i][1] for i in idx], dtype=np.float64)
per = np.diff(starts) / 1e3
med = float(np.median(per))
graph = [k for k in range(len(per)) if per[k] < 1.2 * med]
# the replayed step whose period is closest to the median of the replayed steps (a step picked by POSITION may be one of the
# few that a runtime hiccup stretched - round 4's first pass showed a 402 us step next to a 368 us median)
gmed = float(np.median(per[graph]))
k = min(graph, key=lambda q: abs(per[q] - gmed))
a, b = idx[k], idx[k + 1]
t0, prev, tot = rows[a][1], rows[a - 1][2], 0.0
print("one replayed step (hipGraph), kernel dispatches in stream order")
print("%-46s %10s %9s %8s %6s" % ("kernel", "start_us", "dur_us", "gap_us", "WGs"))
for name, s, e, g, w in rows[a:b]:
    print("%-46s %10.1f %9.1f %8.1f %6d" % (name.split("(")[0][:46], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, g // max(w, 1)))
    prev = e
    tot += (e - s) / 1e3
print("kernels %d, sum of durations %.1f us, step period %.1f us (median of %d replayed steps: %.1f us)"
      % (b - a, tot, per[k], len(graph), float(np.median(per[graph]))))
