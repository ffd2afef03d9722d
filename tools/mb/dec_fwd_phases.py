"""Cycle stamps inside dec_rollout_fwd (library built with -DSW_PHASE_STAMPS: tools/build_variant.sh stamps "-DSW_PHASE_STAMPS",
run with SW_LIB_PATH=variants/lib_stamps.so): cycles per decode step spent in each barrier-delimited phase (wave 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import socialways_amd as sw
from socialways_amd import _lib as L

lib = L.load()
lib.sw_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, To, Tp = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 8, 12
save = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = sw.Generator(use_social=False, device=dev)
G.unify()
obsv, z = torch.rand(B, To, 2, device=dev), torch.rand(B, 32, device=dev)
S = torch.zeros(B, 64, device=dev)
hT, cT = torch.randn(B, 64, device=dev) * 0.1, torch.randn(B, 64, device=dev) * 0.1
pred4 = torch.empty(B, Tp, 4, device=dev)
gsave = torch.zeros(L.workspace_floats(L.WS_GSAVE, B, To, Tp), device=dev)
st = L.stream()
run = lambda: L.call("sw_dec_rollout_fwd", L.ptr(obsv), To, L.ptr(z), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(G.encoder._flat),
                     L.ptr(G.decoder._flat), B, Tp, L.ptr(pred4), None, None, L.ptr(gsave) if save else None, None, 0.0, None, st)
if os.environ.get("SW_GEN_IMAGES", "1") == "1":
    gimg = torch.empty(lib.sw_gen_image_floats(), device=dev)
    L.call("sw_gen_images", L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), None, None, L.ptr(gimg), st)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.sw_debug_stamps(None, 1)
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_longlong * 16)()
lib.sw_debug_stamps(out, 0)
for k, nm in ((13, "prologue: global loads issued"), (14, "prologue: [S|z] tile -> LDS (waits for the loads), barrier"),
              (15, "prologue: u = W1[:, 64:] [S; z] + b1 (72 MFMAs)"), (8, "prologue: barrier")):
    print("%-60s %7.0f cycles per launch (%.2f us)" % (nm, out[k] / N, out[k] / N / 2350.0))
names = {9: "layer 1 (40 MFMAs) + barrier", 10: "layer 2 (52) + barrier", 11: "fc4.fc3 (20) + shuffles", 12: "LSTM step (68) + stores + barrier"}
tot = 0
for k in (9, 10, 11, 12):
    c = out[k] / (N * Tp)
    tot += c
    print("%-40s %7.0f cycles/step" % (names[k], c))
print("sum %.0f cycles/step; kernel %.1f us = %.2f us/step incl. prologue (save=%d)" % (tot, e0.elapsed_time(e1) * 1e3 / N, e0.elapsed_time(e1) * 1e3 / N / Tp, save))
