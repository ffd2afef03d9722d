"""Scratch: per-workgroup start / end stamps of the generator's weight-gradient launch (variant built with -DSW_WG_STAMP)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from socialways_amd import _lib as L
wl = sys.argv[1] if len(sys.argv) > 1 else "m1"
dev = torch.device("cuda:0")
leg = bench.Leg(wl, dev, None, 1, 0, "weak", 256, 4)
leg.tr.use_graph = False
for i in range(4):
    leg.one_step(i)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["SW_LIB_PATH"])
n = 4096
buf = (ctypes.c_ulonglong * (4 * n))()
assert lib.sw_debug_wg_stamps(buf, 4 * n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
if len(sys.argv) > 2 and sys.argv[2] == "d":
    a = a[2048:]
tj = int(a[a[:, 3] > 0][0, 3])
a = a[(a[:, 1] > 0) & (a[:, 3] == tj)]          # the launch grid is padded: workgroups without a job leave no stamp
t00 = a[:, 0].min()
st, en, p = (a[:, 0] - t00) / 100.0, (a[:, 1] - t00) / 100.0, a[:, 2]
print("jobs %d, launch span %.1f us, mean job %.1f us, longest job %.1f us" % (tj, en.max(), (en - st).mean(), (en - st).max()))
for q in np.unique(p):
    m = p == q
    d = en[m] - st[m]
    print("problem %2d: jobs %4d  start %5.1f..%5.1f  dur min %5.1f avg %5.1f max %5.1f  end max %5.1f" % (q, m.sum(), st[m].min(), st[m].max(), d.min(), d.mean(), d.max(), en[m].max()))
d = en - st
idx = np.arange(len(d))
print("per XCD (block %% 8): " + " ".join("%5.1f" % d[idx % 8 == x].mean() for x in range(8)))
print("per XCD max:         " + " ".join("%5.1f" % en[idx % 8 == x].max() for x in range(8)))
