"""Run the generator forward (save + ADE) a few times at m1 size (for rocprofv3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import socialways_amd as sw
from socialways_amd import ops
S, A, To, Tp = 256, 8, 8, 12
t = sw.synth_tracks(S, A, To, Tp, seed=11)
B = S * A
torch.manual_seed(0)
G = sw.Generator(use_social=True, device="cuda:0")
obsv = torch.from_numpy(t["obsvs"]).cuda(); pred = torch.from_numpy(t["preds"]).cuda()
z = torch.rand(B, 32).cuda()
sc = ops.SceneIndex.get(np.asarray(t["batches"]), B, obsv.device)
out = torch.zeros((B + 7) // 8, 3, device="cuda")
ws = ops.Workspaces(obsv.device)
for _ in range(int(os.environ.get("N", "10"))):
    p4, ctx = ops.gen_forward(G.encoder._flat, G.feature_embedder._flat, G.attention._flat, G.decoder._flat, obsv, z, sc, Tp, True,
                              save=True, ws=ws, ade=(pred, 1.0, out))
torch.cuda.synchronize()
