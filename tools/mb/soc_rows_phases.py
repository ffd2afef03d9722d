"""Cycle stamps inside social_pool_bwd_rows_kernel (small scenes, m1 shape).  Library built with -DSW_PHASE_STAMPS."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import socialways_amd as sw
from socialways_amd import _lib as L

lib = L.load()
lib.sw_debug_soc_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
S, A, To, Tp = 256, 8, 8, 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev, use_graph=False)
t = sw.synth_tracks(S, A, To, Tp, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device=dev)
B = S * A
sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
step = lambda: tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, torch.rand(B, 32), data.ss, out=False)
for _ in range(3):
    step()
torch.cuda.synchronize()
lib.sw_debug_soc_stamps(None, 1)
N = 20
for _ in range(N):
    step()
torch.cuda.synchronize()
out = (ctypes.c_longlong * 8)()
lib.sw_debug_soc_stamps(out, 0)
names = {0: "scene prologue (x4, h, Wh, dS, a)", 1: "softmax backward", 2: "weights to registers + the wave's pair tile + row stores issued",
         3: "__syncthreads: row stores drained", 6: "dWh rows (reads f rows back)", 7: "dh rows"}
tot = sum(out[k] for k in names) / N
for k in names:
    c = out[k] / N
    print("%-66s %7.0f cycles (%4.1f %%) %5.2f us" % (names[k], c, 100 * c / tot, c / 2350.0))
print("sum %.0f cycles = %.1f us at 2.35 GHz" % (tot, tot / 2350.0))
