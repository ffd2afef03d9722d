"""Run a few eager m1 training steps (for rocprofv3 counter passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import socialways_amd as sw
S, A = 256, 8
t = sw.synth_tracks(S + 64, A, seed=1234)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
sb = data.the_batches[:S]; B = S * A
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", use_graph=False)
for i in range(int(os.environ.get("N", "6"))):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.02, 0.96, torch.rand(B, 32), data.ss)
torch.cuda.synchronize()
