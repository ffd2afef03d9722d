"""Fixed (prologue / epilogue / launch) vs per-step cost of the serial kernels: time each at two sequence lengths."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import socialways_amd as sw
from socialways_amd import ops, _lib as L
B = int(os.environ.get("B", "2048"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = sw.Generator(use_social=True, device=dev); G.unify()
A = 8
sb = np.stack([np.arange(B // A) * A, (np.arange(B // A) + 1) * A], axis=1).astype(np.int64)
scenes = ops.SceneIndex.get(sb, B, dev)
def timeit(fn, n=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
res = {}
for Tp in (4, 12):
    D = sw.Discriminator(Tp, 64, 2, device=dev)
    obsv = torch.randn(B, 8, 2, device=dev).cumsum(1) * 0.1
    noise = torch.rand(B, 32, device=dev)
    real = torch.randn(B, Tp, 4, device=dev) * 0.1
    ws = ops.Workspaces(dev)
    enc, emb, att, dec = G.encoder, G.feature_embedder, G.attention, G.decoder
    hT, cT, S = torch.randn(B, 64, device=dev) * 0.1, torch.randn(B, 64, device=dev) * 0.1, torch.randn(B, 64, device=dev) * 0.1
    pred4 = torch.empty(B, Tp, 4, device=dev)
    gsave = ws.get("g.gsave", L.workspace_floats(L.WS_GSAVE, B, 8, Tp))
    gdelta = ws.get("g.gdelta", L.workspace_floats(L.WS_GDELTA, B, 8, Tp))
    dh, dc, dS = (torch.empty(B, 64, device=dev) for _ in range(3))
    dpred = torch.randn(B, Tp, 4, device=dev) * 0.01
    def dec_fwd():
        L.call("sw_dec_rollout_fwd", L.ptr(obsv), 8, L.ptr(noise), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(enc._flat), L.ptr(dec._flat), B, Tp,
               L.ptr(pred4), None, None, L.ptr(gsave), None, 0.0, None, L.stream())
    def dec_bwd():
        L.call("sw_dec_rollout_bwd", L.ptr(dpred), L.ptr(enc._flat), L.ptr(dec._flat), L.ptr(gsave), B, 8, Tp, L.ptr(gdelta), L.ptr(dh), L.ptr(dc), L.ptr(dS), L.stream())
    res[("dec_fwd", Tp)] = timeit(dec_fwd)
    res[("dec_bwd", Tp)] = timeit(dec_bwd)
for To in (4, 8):
    D = sw.Discriminator(12, 64, 2, device=dev)
    obsv = torch.randn(B, To, 2, device=dev).cumsum(1) * 0.1
    fake, real = torch.randn(B, 12, 4, device=dev) * 0.1, torch.randn(B, 12, 4, device=dev) * 0.1
    z = torch.rand(B, 32, device=dev); targets = torch.tensor([0.05, 0.95], device=dev)
    ws = ops.Workspaces(dev)
    g = torch.zeros_like(D._flat); part = torch.zeros((B + 7) // 8, 3, device=dev)
    st = {}
    def d_fwd():
        st["o"] = ops.disc_forward(D._flat, obsv, [fake, real], save=True, ws=ws)
    d_fwd()
    def d_bwd_nowg():   # data gradients only is not separable: time the whole call, then subtract the wgrad kernels below
        labels, codes, ctx = st["o"]
        ops.disc_backward_gan(D._flat, ctx, labels, codes, targets, (0, 1), z, 1.0 / B, 0.25 / B, g, (), ws=ws, loss_part=part)
    def d_pred():
        ops.disc_dpred(D._flat, obsv, fake, targets, 1, z, 1.0 / B, 0.25 / B)
    res[("disc_fwd", To)] = timeit(d_fwd)
    res[("disc_bwd+wgrad", To)] = timeit(d_bwd_nowg)
    res[("disc_dpred", To)] = timeit(d_pred)
def report(name, a, b):
    ta, tb = res[(name, a)], res[(name, b)]
    per = (tb - ta) / (b - a)
    print("%-16s T=%d %.1f us, T=%d %.1f us -> %.2f us/step, fixed %.1f us" % (name, a, ta, b, tb, per, tb - b * per))
report("dec_fwd", 4, 12); report("dec_bwd", 4, 12); report("disc_fwd", 4, 8); report("disc_bwd+wgrad", 4, 8); report("disc_dpred", 4, 8)
