"""Cycle stamps inside disc_fwd_kernel / disc_bwd_kernel over whole training steps at m1 (library built with
-DSW_PHASE_STAMPS): cycles of thread 0 of workgroup 0, summed over the launches of a step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import socialways_amd as sw
from socialways_amd import _lib as L

lib = L.load()
lib.sw_debug_disc_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
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
lib.sw_debug_disc_stamps(None, 1)
N = 20
for _ in range(N):
    step()
torch.cuda.synchronize()
out = (ctypes.c_longlong * 16)()
lib.sw_debug_disc_stamps(out, 0)
names = {0: "fwd: prologue (3 launches: heads-only pass 1, pass 2, generator phase)", 1: "fwd: observation LSTM (2 launches x 8 steps)",
         2: "fwd: observation fc", 3: "fwd: heads (1 branch per workgroup / the G-phase branch)", 4: "fwd: fused heads backward (G phase)",
         8: "bwd: prologue (2 launches)", 9: "bwd: heads, 2 branches (2 launches)", 10: "bwd: observation fc + first rows",
         11: "bwd: BPTT (2 launches x 8 steps) - with finer stamps: loop exit only",
         12: "bwd: BPTT: row loads issued, cell backward, dgates -> LDS / HBM", 13: "bwd: BPTT: barrier", 14: "bwd: BPTT: dh_prev = W_hh^T dgates (64 MFMA)"}
for k in names:
    c = out[k] / N
    print("%-76s %8.0f cycles/step %6.2f us" % (names[k], c, c / 2350.0))
