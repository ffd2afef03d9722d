"""Randomised epoch-level sweep: train_epoch() + test() of the HIP trainers against the CPU oracle on random RAGGED
datasets (the reference's greedy scene packing, train.py:446-456; K sampled futures per held-out scene, train.py:563-616)
with the same RNG streams - epoch ADE / FDE, every step's MSE terms, the packed-batch sizes, min / avg ADE and FDE.
python tools/dbg/fuzz_epoch.py [n] [seed]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
import sw_oracle as O


def run(N=10, seed=0):
    rng = np.random.default_rng(seed)
    fails = 0
    for it in range(N):
        amax = int(rng.choice([2, 8, 8, 20, 64]))
        n_sc = int(rng.choice([12, 40, 90]))
        sizes = [int(rng.integers(1, amax + 1)) for _ in range(n_sc)]
        To, Tp = int(rng.choice([3, 8, 8])), int(rng.choice([2, 8, 12]))
        H = int(rng.choice([64, 64, 32, 128, 80]))
        bs = int(rng.choice([16, 64, 256]))
        K = int(rng.choice([1, 3, 20]))
        kw = dict(use_social=bool(rng.random() < 0.85), n_unrolling_steps=int(rng.choice([0, 1, 2])))
        s_t, s_w, s_r = (int(rng.integers(1 << 30)) for _ in range(3))
        cfg = dict(sizes="%d scenes, %d agents, max %d" % (n_sc, sum(sizes), max(sizes)), To=To, Tp=Tp, H=H, bs=bs, K=K, **kw)
        t0 = time.perf_counter()
        try:
            t = sw.synth_tracks(n_sc, sizes, To, Tp, seed=s_t)
            data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
            odata = O.load_and_normalise(t["obsvs"], t["preds"], t["batches"])
            torch.manual_seed(s_w)
            tr = sw.SocialWaysTrainer(Tp, hidden_size=H, device="cuda:0", **kw)
            orc = O.SocialWaysOracle(Tp, hidden_size=H, **kw)
            orc.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
            errs = []
            for ep in range(2):
                torch.manual_seed(s_r + ep); np.random.seed(s_r + ep)
                ade, fde, losses, szs = tr.train_epoch(data, bs)
                torch.manual_seed(s_r + ep); np.random.seed(s_r + ep)
                o = orc.train_epoch(odata, bs)
                oade, ofde, olosses = o[0], o[1], np.asarray(o[2])
                osz = o[3] if len(o) > 3 else None
                l = np.asarray(losses)
                keep = [i for i in range(l.shape[1]) if i != l.shape[1] - 3]
                e_l = float(np.max(np.abs(l[:, keep] - olosses[:, keep]) / (np.abs(olosses[:, keep]) + 1e-5)))
                errs.append((abs(ade - oade) / max(oade, 1e-9), abs(fde - ofde) / max(ofde, 1e-9), e_l))
                assert osz is None or [tuple(x) for x in szs] == [tuple(x) for x in osz], "packed-batch sizes differ"
            torch.manual_seed(s_r + 9)
            got = np.asarray(tr.test(data, n_gen_samples=K))
            torch.manual_seed(s_r + 9)
            want = np.asarray(orc.test(odata, n_gen_samples=K))
            e_t = float(np.max(np.abs(got - want) / (np.abs(want) + 1e-9)))
            # epoch 0 runs on identical weights; epoch 1 and test() follow weights that differ by Adam's sign noise (~lr)
            ok = errs[0][0] < 1e-4 and errs[0][1] < 1e-4 and errs[0][2] < 2e-2 and errs[1][0] < 5e-2 and e_t < 5e-2 \
                and np.isfinite(got).all()
            print("%s #%02d %-14s %s | epoch0 ade %.1e fde %.1e loss %.1e | epoch1 ade %.1e | test %.1e | %.1fs"
                  % ("ok  " if ok else "FAIL", it, type(tr).__name__, cfg, *errs[0], errs[1][0], e_t, time.perf_counter() - t0), flush=True)
            fails += 0 if ok else 1
        except Exception as e:
            if isinstance(e, sw.SocialWaysHipError) and "not supported" in str(e):
                print("skip #%02d %s: %s" % (it, cfg, str(e)[:90]), flush=True)
                continue
            fails += 1
            print("EXC  #%02d %s\n%s" % (it, cfg, traceback.format_exc()), flush=True)
    print("%d configurations, %d failures" % (N, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
