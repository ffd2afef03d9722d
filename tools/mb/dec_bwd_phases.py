"""Cycle stamps inside dec_rollout_bwd (library built with -DSW_PHASE_STAMPS: tools/build_variant.sh stamps "-DSW_PHASE_STAMPS",
run with SW_LIB_PATH=variants/lib_stamps.so): cycles per decode step spent in each barrier-delimited phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import socialways_amd as sw
from socialways_amd import _lib as L

lib = L.load()
lib.sw_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, To, Tp = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 8, 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = sw.Generator(use_social=False, device=dev)
G.unify()
obsv, z = torch.rand(B, To, 2, device=dev), torch.rand(B, 32, device=dev)
S = torch.zeros(B, 64, device=dev)
hT, cT = torch.randn(B, 64, device=dev) * 0.1, torch.randn(B, 64, device=dev) * 0.1
pred4 = torch.empty(B, Tp, 4, device=dev)
gsave = torch.zeros(L.workspace_floats(L.WS_GSAVE, B, To, Tp), device=dev)
gdelta = torch.empty(L.workspace_floats(L.WS_GDELTA, B, To, Tp), device=dev)
st = L.stream()
L.call("sw_dec_rollout_fwd", L.ptr(obsv), To, L.ptr(z), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(G.encoder._flat), L.ptr(G.decoder._flat),
       B, Tp, L.ptr(pred4), None, None, L.ptr(gsave), None, 0.0, None, st)
dp = torch.randn(B, Tp, 4, device=dev)
dh, dc, dS = (torch.empty(B, 64, device=dev) for _ in range(3))
run = lambda: L.call("sw_dec_rollout_bwd", L.ptr(dp), L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), L.ptr(gsave), B, To, Tp,
                     L.ptr(gdelta), L.ptr(dh), L.ptr(dc), L.ptr(dS), st)
if os.environ.get("SW_GEN_IMAGES", "1") == "1":      # the trainer's default: weight images derived once per step
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
pro = {5: "prologue: W_hh^T loads issued, LDS zeroed, barrier", 6: "prologue: weight staging + input-matrix rows, barrier",
       2: "prologue: W43^T / Wx^T composition, barrier"}
names = {7: "P6 dh += W1h^T dz1 (+ loop top, loads issue)", 0: "P1 cell_bwd, dgates -> LDS/HBM, barrier", 1: "P2 dh_prev + dx4 partial, barrier",
          3: "P3+P4 dx4 sum, dv, dz2 (composed fc4.fc3), barrier", 4: "P5 dz1, barrier"}
for k in (5, 6, 2):
    print("%-60s %7.0f cycles per launch (%.2f us)" % (pro[k], out[k] / N, out[k] / N / 2350.0))
tot = 0
for k in (0, 1, 3, 4, 7):
    c = out[k] / (N * Tp)
    tot += c
    print("%-52s %7.0f cycles/step" % (names[k], c))
print("sum %.0f cycles/step; kernel %.1f us = %.2f us/step incl. prologue" % (tot, e0.elapsed_time(e1) * 1e3 / N, e0.elapsed_time(e1) * 1e3 / N / Tp))
