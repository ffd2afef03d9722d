"""Driver for rocprofv3: N graph-replayed wide-path steps at the metric shape.  python tools/dbg/wide_prof.py [H] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import socialways_amd as sw
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(12, hidden_size=H, device="cuda:0")
S, A = 256, 8
t = sw.synth_tracks(S, A, 8, 12, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
B, sb = S * A, data.the_batches[:S]
noise = torch.rand(B, H // 2)
for _ in range(4):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%s H=%d: %.3f ms/step = %.1f steps/s" % (type(tr).__name__, H, 1e3 * dt, 1 / dt))
