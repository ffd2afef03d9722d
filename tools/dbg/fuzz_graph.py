"""Randomised state sweep: a trainer that captures / replays hipGraphs against the same trainer running every step eagerly, on
a RANDOM SEQUENCE of packed-batch layouts (ragged scenes, batch sizes that grow and shrink, layouts that recur so that they get
captured, more layouts than the caches hold, K-step launches) - the reported sums and every weight must stay bit-identical.
python tools/dbg/fuzz_graph.py [hidden] [steps] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import socialways_amd as sw


def run(H=64, steps=120, seed=0, n_layouts=14):
    rng = np.random.default_rng(seed)
    sizes = [int(rng.integers(1, int(rng.choice([3, 8, 20, 64])) + 1)) for _ in range(400)]
    t = sw.synth_tracks(len(sizes), sizes, 8, 12, seed=int(rng.integers(1 << 30)))
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    tb = data.the_batches
    layouts = []
    for _ in range(n_layouts):                    # a layout = a run of consecutive scenes
        s0 = int(rng.integers(0, len(sizes) - 60))
        n = int(rng.integers(1, 60))
        layouts.append((s0, n))
    kw = dict(use_social=True, n_unrolling_steps=int(rng.choice([0, 1, 2])))
    wseed = int(rng.integers(1 << 30))
    torch.manual_seed(wseed)
    a = sw.SocialWaysTrainer(12, hidden_size=H, device="cuda:0", use_graph=True, **kw)
    torch.manual_seed(wseed)
    b = sw.SocialWaysTrainer(12, hidden_size=H, device="cuda:0", use_graph=False, **kw)
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    bad = 0
    i = 0
    while i < steps:
        s0, n = layouts[int(rng.integers(0, len(layouts) if rng.random() < 0.7 else min(4, len(layouts))))]
        r0, r1 = int(tb[s0][0]), int(tb[s0 + n - 1][1])
        sb = np.asarray(tb[s0:s0 + n], dtype=np.int64) - r0
        B = r1 - r0
        k = int(rng.choice([1, 1, 1, 2, 4]))
        batches = [(data.obsv[r0:r1], data.pred[r0:r1], float(rng.uniform(0, 0.1)), float(rng.uniform(0.9, 1.0)),
                    torch.rand(B, H // 2, generator=gen)) for _ in range(k)]
        if k == 1:
            ra = [a.step(*batches[0][:2], sb, *batches[0][2:], data.ss)]
            rb = [b.step(*batches[0][:2], sb, *batches[0][2:], data.ss)]
        else:
            ra = a.step_many(batches, sb, data.ss)
            rb = b.step_many(batches, sb, data.ss)
        same = all(torch.equal(x, y) for x, y in zip(ra, rb))
        wa = torch.cat([p.detach().reshape(-1) for p in list(a.G.parameters()) + list(a.D.parameters())])
        wb = torch.cat([p.detach().reshape(-1) for p in list(b.G.parameters()) + list(b.D.parameters())])
        same = same and torch.equal(wa, wb) and bool(torch.isfinite(wa).all())
        if not same:
            bad += 1
            print("MISMATCH at step %d: layout (%d scenes from %d, B = %d), k = %d, sums equal %s, max |dW| %.3e"
                  % (i, n, s0, B, k, all(torch.equal(x, y) for x, y in zip(ra, rb)), float((wa - wb).abs().max())), flush=True)
            if bad > 3:
                break
        i += k
    print("%s H=%d U=%d: %d steps over %d layouts, %d mismatches, %d layouts captured"
          % (type(a).__name__, H, kw["n_unrolling_steps"], i, len(layouts), bad, len(getattr(a, "_graphs", {}))), flush=True)
    return bad


if __name__ == "__main__":
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = run(H, int(sys.argv[2]) if len(sys.argv) > 2 else 120, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    sys.exit(1 if n else 0)
