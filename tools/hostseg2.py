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
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 60
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
t_all = time.perf_counter()
for i in range(N):
    t = time.perf_counter()
    nz = torch.rand(B, 32) if mode == "rand" else noise
    tick("rand", t)
    t = time.perf_counter(); st["obsv"].copy_(data.obsv[:B]); st["pred"].copy_(data.pred[:B]); tick("d2d in", t)
    k = i % 4
    host, devslot, ready, consumed = st["ring"][k]
    t = time.perf_counter(); ready.synchronize(); tick("ready.sync", t)
    t = time.perf_counter(); host[0], host[1] = 0.05, 0.95; np.copyto(host[2:].view(B, 32).numpy(), nz.numpy()); tick("memcpy", t)
    main = torch.cuda.current_stream()
    t = time.perf_counter()
    with torch.cuda.stream(st["copy_stream"]):
        st["copy_stream"].wait_event(consumed)
        devslot.copy_(host, non_blocking=True)
        ready.record(st["copy_stream"])
    tick("h2d(copy stream)", t)
    t = time.perf_counter(); main.wait_event(ready); st["targets"].copy_(devslot[:2]); st["noise"].copy_(devslot[2:].view(B, 32)); consumed.record(main); tick("d2d noise", t)
    t = time.perf_counter()
    for g, buf in st["graph"]:
        g.replay()
    tick("replay", t)
    t = time.perf_counter(); out.copy_(st["out"]); tick("out copy", t)
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print("mode", mode, "wall/step %.1f us" % (wall / N * 1e6))
for k, v in T.items():
    print("  %-18s %8.1f us/step" % (k, v / N * 1e6))
