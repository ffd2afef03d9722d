"""Soak run: many epochs of train_epoch() on a RAGGED synthetic crowd (hundreds of distinct packed-batch layouts: graph
capture, layout caches, workspace eviction) - losses stay finite, device memory stops growing.
python tools/dbg/soak.py [hidden] [epochs] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import socialways_amd as sw
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
E = int(sys.argv[2]) if len(sys.argv) > 2 else 30
BS = int(sys.argv[3]) if len(sys.argv) > 3 else 256
torch.manual_seed(0); np.random.seed(0)
sizes = sw.ragged_scene_sizes(n_agents=6000, max_agents=8, seed=5)
t = sw.synth_tracks(len(sizes), list(sizes), 8, 12, seed=2)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
tr = sw.SocialWaysTrainer(12, hidden_size=H, device="cuda:0")
print(type(tr).__name__, "scenes", len(sizes), "train samples", data.n_train_samples)
mem = []
t0 = time.perf_counter()
for e in range(E):
    ade, fde, losses, szs = tr.train_epoch(data, BS)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    mem.append(total - free)
    l = np.asarray(losses)
    assert np.isfinite(l).all() and np.isfinite(ade) and np.isfinite(fde), (e, ade, fde)
    if e % max(1, E // 10) == 0 or e == E - 1:
        print("epoch %3d: ADE %.4f FDE %.4f d_loss %.4f g_loss %.4f steps %d layouts-in-cache %d device MB %.0f (%.1f s)"
              % (e, ade, fde, l[:, :3].sum(1).mean(), l[:, 7:].sum(1).mean() if l.shape[1] > 8 else l[:, -2:].sum(1).mean(),
                 len(szs), len(getattr(tr, "_graphs", {})), mem[-1] / 2**20, time.perf_counter() - t0))
half = len(mem) // 2
print("device memory: epoch 0 %.0f MB, mid %.0f MB, last %.0f MB -> growth over the second half %.1f MB"
      % (mem[0] / 2**20, mem[half] / 2**20, mem[-1] / 2**20, (mem[-1] - mem[half]) / 2**20))
res = tr.test(data, n_gen_samples=20)
print("test():", res)
