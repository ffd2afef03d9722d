"""Host-vs-GPU time per training step: wall clock, per-step GPU span (events), host enqueue time."""
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
def one(i, noise=None):
    a = (i % 8) * B
    if noise is None:
        noise = torch.rand(B, 32)
    tr.step(data.obsv[a:a + B], data.pred[a:a + B], sb, 0.05, 0.95, noise, data.ss, out=out)
for i in range(10): one(i)
torch.cuda.synchronize()
for mode in ("host rand", "fixed noise"):
    fixed = torch.rand(B, 32) if mode == "fixed noise" else None
    evs = []; host = 0.0
    t0 = time.perf_counter()
    for i in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record(); one(i, fixed); e1.record()
        host += time.perf_counter() - h0
        evs.append((e0, e1))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu = np.array([a.elapsed_time(b) for a, b in evs])
    print("%-12s wall/step %.1f us | host enqueue/step %.1f us | GPU span/step mean %.1f min %.1f max %.1f us"
          % (mode, wall / 50 * 1e6, host / 50 * 1e6, gpu.mean() * 1e3, gpu.min() * 1e3, gpu.max() * 1e3))
