"""Debug aid: steps/s of the default trainer at S scenes x A agents (8 + 12 steps), graph replay.
python tools/dbg/step_time.py S A [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import socialways_amd as sw
S, A = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(12, device="cuda:0")
t = sw.synth_tracks(S, A, 8, 12, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
B, sb = S * A, data.the_batches[:S]
noise = torch.rand(B, 32)
best = 1e9
for rep in range(4):
    for _ in range(6 if rep == 0 else 0):
        tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
print("S=%d A=%d (%d tiles): %.4f ms/step = %.1f steps/s" % (S, A, (B + 15) // 16, 1e3 * best, 1 / best))
