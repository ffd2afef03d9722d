"""Pure host cost of one graph-replayed step (GPU idle before each call => no back-pressure waits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import socialways_amd as sw
dev = torch.device("cuda:0")
S, A, To, Tp = 256, 8, 8, 12
B = S * A
torch.manual_seed(0); np.random.seed(0)
tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev)
tracks = sw.synth_tracks(S * 8, A, To, Tp, seed=1234)
data = sw.SceneDataset(tracks["obsvs"], tracks["preds"], tracks["batches"], device=dev)
sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
out = torch.zeros(4, 3, device=dev)
noise = torch.rand(B, 32)
for i in range(8):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, noise, data.ss, out=out)
torch.cuda.synchronize()
import cProfile, pstats
ts = []
for i in range(30):
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, noise, data.ss, out=out)
    ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
print("host cost of step() with idle GPU: mean %.1f us  min %.1f us" % (np.mean(ts) * 1e6, np.min(ts) * 1e6))
t = time.perf_counter()
for i in range(30):
    torch.rand(B, 32)
print("torch.rand(B,32): %.1f us" % ((time.perf_counter() - t) / 30 * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    torch.cuda.synchronize()
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, noise, data.ss, out=out)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(14)
