"""Randomised checkpoint sweep: train a few epochs, save (the reference's 8-key dict, train.py:651-663), load into a FRESH
trainer (different seed), continue both: the continued run must equal the uninterrupted one bit for bit (weights, Adam moments
and step counts all travel through the file); the file must load with torch.load on the CPU and carry the reference's keys /
shapes.  python tools/dbg/fuzz_resume.py [n] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
import sw_oracle as O


def run(N=10, seed=0):
    rng = np.random.default_rng(seed)
    fails = 0
    for it in range(N):
        sizes = [int(rng.integers(1, int(rng.choice([3, 8, 20])) + 1)) for _ in range(int(rng.choice([10, 40])))]
        To, Tp = int(rng.choice([3, 8])), int(rng.choice([2, 12]))
        H = int(rng.choice([64, 32, 128, 80, 16]))
        nl = 2 if H <= 64 else int(rng.choice([2, 3]))
        kw = dict(use_social=bool(rng.random() < 0.85), n_unrolling_steps=int(rng.choice([0, 1, 2])), n_latent_codes=nl)
        bs = int(rng.choice([16, 64]))
        s_t, s_w, s_r = (int(rng.integers(1 << 30)) for _ in range(3))
        t = sw.synth_tracks(len(sizes), sizes, To, Tp, seed=s_t)
        data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
        torch.manual_seed(s_w)
        a = sw.SocialWaysTrainer(Tp, hidden_size=H, device="cuda:0", **kw)

        def epochs(tr, e0, n):
            out = []
            for ep in range(e0, e0 + n):
                torch.manual_seed(s_r + ep); np.random.seed(s_r + ep)
                out.append(tr.train_epoch(data, bs)[:3])
            return out
        epochs(a, 0, 2)
        path = "/tmp/fuzz_resume_%d.pt" % os.getpid()
        a.save(path, epoch=2)
        ck = torch.load(path, map_location="cpu")
        keys_ok = sorted(ck.keys()) == sorted(['epoch', 'attentioner_dict', 'feature_embedder_dict', 'encoder_dict',
                                               'decoder_dict', 'pred_optimizer', 'D_dict', 'D_optimizer'])
        # the oracle (the reference's module definitions at this width) must accept the weights as they are
        orc = O.SocialWaysOracle(Tp, hidden_size=H, **kw)
        try:
            orc.load_state(ck)
            shapes_ok = True
        except Exception as e:
            shapes_ok = False
            print("   oracle refuses the checkpoint:", str(e)[:200])
        torch.manual_seed(s_w + 1)
        b = sw.SocialWaysTrainer(Tp, hidden_size=H, device="cuda:0", **kw)
        nxt = b.load_checkpoint(path)
        ra, rb = epochs(a, 2, 2), epochs(b, 2, 2)
        same = nxt == 3 and all(x[0] == y[0] and x[1] == y[1] and np.array_equal(np.asarray(x[2]), np.asarray(y[2])) for x, y in zip(ra, rb))
        for (k, p), (_, q) in zip(list(a.G.state_dict().items()) + list(a.D.state_dict().items()),
                                  list(b.G.state_dict().items()) + list(b.D.state_dict().items())):
            same = same and torch.equal(p, q)
        ok = same and keys_ok and shapes_ok
        print("%s #%02d %s H=%d nl=%d To=%d Tp=%d %s: resumed == uninterrupted %s, reference keys %s, reference shapes %s"
              % ("ok  " if ok else "FAIL", it, type(a).__name__, H, nl, To, Tp, {k: v for k, v in kw.items() if k != "n_latent_codes"},
                 same, keys_ok, shapes_ok), flush=True)
        fails += 0 if ok else 1
    print("%d configurations, %d failures" % (N, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
