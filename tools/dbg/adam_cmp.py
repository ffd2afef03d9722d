import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import socialways_amd as sw
t = sw.synth_tracks(24, 8, seed=9)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
B, sb = data.n_train_samples, data.train_batches
res = []
for fuse in (True, False):
    torch.manual_seed(0)
    tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", use_graph=False)
    tr._fuse_d_adam = fuse
    gen = torch.Generator().manual_seed(5)
    for i in range(3):
        tr.step(data.obsv[:B], data.pred[:B], sb, 0.01 * i, 0.9, torch.rand(B, 32, generator=gen), data.ss)
    res.append((tr.D._flat.clone(), tr.D_optimizer.m.clone(), tr.D_optimizer.v.clone()))
for name, a, b in zip(("w", "m", "v"), res[0], res[1]):
    d = (a - b).abs()
    nz = (d > 0).sum().item()
    print(name, "max abs diff %.3e, differing elements %d of %d, max rel %.3e" % (d.max().item(), nz, d.numel(), (d / b.abs().clamp_min(1e-30)).max().item()))
