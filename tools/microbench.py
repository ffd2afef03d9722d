"""Per-kernel timing through the C ABI with HIP events: separates prologue from per-step cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import socialways_amd as sw
from socialways_amd import _lib as L, ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
G = sw.Generator(use_social=True, device=dev)
G.unify()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


st = L.stream()
for T in (2, 4, 8, 16):
    obsv = torch.rand(B, T, 2, device=dev)
    hT, cT = torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev)
    act = torch.empty(T * B * 384, device=dev)
    x4s = torch.empty(T * B * 4, device=dev)
    f = lambda: L.call("sw_enc_lstm_fwd", L.ptr(obsv), 0, L.ptr(G.encoder._flat), None, None, B, T, L.ptr(hT), L.ptr(cT), None, L.ptr(act), L.ptr(x4s), 0, st)
    f2 = lambda: L.call("sw_enc_lstm_fwd", L.ptr(obsv), 0, L.ptr(G.encoder._flat), None, None, B, T, L.ptr(hT), L.ptr(cT), None, None, None, 0, st)
    print("enc_lstm_fwd T=%2d  save %7.1f us   nosave %7.1f us" % (T, timeit(f), timeit(f2)))
obsv = torch.rand(B, 8, 2, device=dev)
z = torch.rand(B, 32, device=dev)
S = torch.zeros(B, 64, device=dev)
hT, cT = torch.randn(B, 64, device=dev) * 0.1, torch.randn(B, 64, device=dev) * 0.1
for Tp in (1, 2, 4, 12):
    pred4 = torch.empty(B, Tp, 4, device=dev)
    gsave = torch.empty(L.workspace_floats(L.WS_GSAVE, B, 8, Tp), device=dev)
    f = lambda: L.call("sw_dec_rollout_fwd", L.ptr(obsv), 8, L.ptr(z), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), B, Tp, L.ptr(pred4), None, None, L.ptr(gsave), st)
    f2 = lambda: L.call("sw_dec_rollout_fwd", L.ptr(obsv), 8, L.ptr(z), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), B, Tp, L.ptr(pred4), None, None, None, st)
    t1, t2 = timeit(f), timeit(f2)
    gdelta = torch.empty(L.workspace_floats(L.WS_GDELTA, B, 8, Tp), device=dev)
    dp = torch.randn(B, Tp, 4, device=dev)
    dh, dc, dS = torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev)
    fb = lambda: L.call("sw_dec_rollout_bwd", L.ptr(dp), L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), L.ptr(gsave), B, 8, Tp, L.ptr(gdelta), L.ptr(dh), L.ptr(dc), L.ptr(dS), st)
    print("dec_rollout Tp=%2d  fwd save %7.1f us  nosave %7.1f us   bwd %7.1f us" % (Tp, t1, t2, timeit(fb)))
D = sw.Discriminator(12, 64, 2, device=dev)
for To in (2, 8, 16):
    obsv = torch.rand(B, To, 2, device=dev)
    p1, p2 = torch.randn(B, 12, 4, device=dev), torch.randn(B, 12, 4, device=dev)
    f1 = lambda: ops.disc_forward(D._flat, obsv, [p1], save=False)
    f2 = lambda: ops.disc_forward(D._flat, obsv, [p1, p2], save=False)
    print("disc_fwd To=%2d  nb=1 %7.1f us   nb=2 %7.1f us" % (To, timeit(f1), timeit(f2)))
