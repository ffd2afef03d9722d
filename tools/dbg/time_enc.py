"""Time sw_enc_lstm_fwd / bwd stand-alone for T = 8 and T = 40 in both tile modes: per-step cost and fixed cost."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import socialways_amd as sw
from socialways_amd import _lib as L
B = int(os.environ.get("B", "2048"))
torch.manual_seed(0)
G = sw.Generator(use_social=True, device="cuda:0")
enc = G.encoder._flat
lib = L.load()
for mode in (1, 2):
    lib.sw_set_tile_mode(mode)
    res = {}
    for T in (8, 40):
        x = torch.rand(B, T, 2, device="cuda")
        hT, cT = torch.empty(B, 64, device="cuda"), torch.empty(B, 64, device="cuda")
        act = torch.empty(T * B * 384, device="cuda"); x4s = torch.empty(T * B * 4, device="cuda")
        dg = torch.empty(T * B * 256, device="cuda"); dh = torch.randn(B, 64, device="cuda"); dc = torch.randn(B, 64, device="cuda")
        def fwd():
            L.call("sw_enc_lstm_fwd", L.ptr(x), 0, L.ptr(enc), None, None, B, T, L.ptr(hT), L.ptr(cT), None, L.ptr(act), L.ptr(x4s), 0, L.stream())
        def bwd():
            L.call("sw_enc_lstm_bwd", L.ptr(enc), L.ptr(act), None, L.ptr(dh), L.ptr(dc), None, B, T, 0, L.ptr(dg), None, None, L.stream())
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            res[(name, T)] = e0.elapsed_time(e1) * 1e3 / n
    for name in ("fwd", "bwd"):
        per = (res[(name, 40)] - res[(name, 8)]) / 32
        print("mode %d %s: T=8 %.1f us, T=40 %.1f us -> %.2f us/step, fixed %.1f us" % (mode, name, res[(name, 8)], res[(name, 40)], per, res[(name, 8)] - 8 * per))
lib.sw_set_tile_mode(0)
