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
for i in range(6):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, noise, data.ss, out=out)
torch.cuda.synchronize()
st = list(tr._graphs.values())[0]
g = st["graph"]
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 50
for i in range(N):
    t = time.perf_counter(); st["obsv"].copy_(data.obsv[:B]); st["pred"].copy_(data.pred[:B]); tick("d2d copies", t)
    t = time.perf_counter(); slot, ev = st["ring"][i % 4]; ev.synchronize(); tick("ev.sync", t)
    t = time.perf_counter(); np.copyto(slot[2:].view(B, 32).numpy(), noise.numpy()); tick("memcpy", t)
    t = time.perf_counter(); st["noise"].copy_(slot[2:].view(B, 32), non_blocking=True); st["targets"].copy_(slot[:2], non_blocking=True); ev.record(); tick("h2d", t)
    t = time.perf_counter(); g.replay(); tick("replay", t)
    t = time.perf_counter(); out.copy_(st["out"]); tick("out copy", t)
torch.cuda.synchronize()
for k, v in T.items():
    print("%-12s %8.1f us/step" % (k, v / N * 1e6))
