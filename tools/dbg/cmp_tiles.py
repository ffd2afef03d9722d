"""Debug: 8-agent vs 16-agent tiling of the generator forward (rollout, saved rows, ADE sums) on an odd batch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import socialways_amd as sw
from socialways_amd import ops, _lib as L
sizes = [1, 5, 16, 2, 13]; To, Tp = int(os.environ.get("TO", 3)), int(os.environ.get("TP", 5))
t = sw.synth_tracks(len(sizes), sizes, To, Tp, seed=11)
B = sum(sizes)
torch.manual_seed(0)
G = sw.Generator(use_social=True, device="cuda:0")
obsv = torch.from_numpy(t["obsvs"]).cuda(); pred = torch.from_numpy(t["preds"]).cuda()
z = torch.rand(B, 32).cuda()
sc = ops.SceneIndex.get(np.asarray(t["batches"]), B, obsv.device)
res = {}
for mode in (1, 2):
    L.load().sw_set_tile_mode(mode)
    out = torch.zeros((B + 7) // 8, 3, device="cuda")
    ws = ops.Workspaces(obsv.device)
    p4, ctx = ops.gen_forward(G.encoder._flat, G.feature_embedder._flat, G.attention._flat, G.decoder._flat, obsv, z, sc, Tp, True,
                              save=True, ws=ws, ade=(pred, 1.0, out))
    torch.cuda.synchronize()
    gs = ctx.gsave.clone()
    res[mode] = (p4.clone(), out.sum(0).clone(), gs, ctx.hT.clone(), ctx.S.clone())
L.load().sw_set_tile_mode(0)
a, b = res[1], res[2]
print("pred4 max diff", (a[0] - b[0]).abs().max().item(), "hT", (a[3] - b[3]).abs().max().item(), "S", (a[4]-b[4]).abs().max().item())
print("ade sums wide", a[1].tolist(), "narrow", b[1].tolist())
Ta = To + Tp - 1
n_act = Ta * B * 384; n_x = Ta * B * 4; n_a1 = Tp * B * 160; n_a2 = Tp * B * 80
off = 0
for name, n in (("act", n_act), ("x4s", n_x), ("a1", n_a1), ("a2", n_a2)):
    d = (a[2][off:off + n] - b[2][off:off + n]).abs()
    print(name, "max diff", d.max().item(), "at", int(d.argmax()))
    off += n
err = ((a[0][:, :, :2] - pred) ** 2).sum(2)
print("direct: sum err/Tp %.6f last %.6f sq %.6f" % (err.sqrt().sum().item() / Tp, err.sqrt()[:, -1].sum().item(), err.sum().item()))
