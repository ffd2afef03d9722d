"""Cycle stamps inside social_pool_bwd_kernel (dense scenes).  Library built with -DSW_PHASE_STAMPS
(bash tools/build_variant.sh stamps "-DSW_PHASE_STAMPS"), run with SW_LIB_PATH=variants/lib_stamps.so:
cycles of wave 0 of workgroup 0 (one 64-agent scene per launch at c4) by phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import socialways_amd as sw
from socialways_amd import _lib as L

lib = L.load()
lib.sw_debug_soc_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
S, A, To, Tp = 512, 64, 8, 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev, use_graph=False)
t = sw.synth_tracks(S, A, To, Tp, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device=dev)
B = S * A
sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
step = lambda: tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, torch.rand(B, 32), data.ss, out=False)
for _ in range(2):
    step()
torch.cuda.synchronize()
lib.sw_debug_soc_stamps(None, 1)
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
out = (ctypes.c_longlong * 8)()
lib.sw_debug_soc_stamps(out, 0)
names = {0: "scene prologue (x4, h, Wh, dS, a)", 1: "softmax backward",
         2: "tile: features, fc.0, fc.2 forward, Q_j / dh2 on the VALU (no fc.4 per pair)",
         3: "tile: (was dW3 + dh2: 128 MFMAs per tile - now per j block)", 4: "tile: dW2 (32), dh1 (32)", 5: "tile: dW1 (VALU)",
         6: "per j-block: v_j, dW3 += Wh Q^T, dWh_j = W3 Q_j + b3 sd_j", 7: "dh rows"}
tiles = 64 * 4 / 4.0      # tiles of wave 0 per scene
tot = sum(out[k] for k in range(8)) / N
for k in range(8):
    c = out[k] / N
    per = "  = %6.0f cycles per tile" % (c / tiles) if k in (2, 3, 4, 5) else ""
    print("%-58s %9.0f cycles per scene (%4.1f %%)%s" % (names[k], c, 100 * c / tot, per))
print("sum %.0f cycles per scene = %.1f us at 2.35 GHz" % (tot, tot / 2350.0))
