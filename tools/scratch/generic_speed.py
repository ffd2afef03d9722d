"""Scratch: steps/s of the generic-width path (hidden size 128) at the metric shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import socialways_amd as sw
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S, A, To, Tp = 256, 8, 8, 12
dev = torch.device("cuda:0")
tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev, hidden_size=H)
t = sw.synth_tracks(S, A, To, Tp, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device=dev)
B = S * A
sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
step = lambda: tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, torch.rand(B, H // 2), data.ss)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("hidden size %d (%s): %.2f steps/s, %.1f ms per step" % (H, type(tr).__name__, n / dt, 1e3 * dt / n))
