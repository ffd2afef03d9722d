"""Randomised sweep of the MODULE surface: the reference's training step written literally with the package's modules and
torch autograd - three predict() calls, separate D calls, nn.MSELoss, deepcopy / D.load, torch.optim.Adam (train.py:470-543) -
on random ragged batches (scenes above 64 agents included) and hidden sizes, against the CPU oracle: the 9 MSE terms, every
gradient of G's last backward, every weight after the step.  python tools/dbg/fuzz_literal.py [n] [seed]"""
import copy, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
import sw_oracle as O


def run(N=20, seed=0):
    rng = np.random.default_rng(seed)
    fails = 0
    for it in range(N):
        amax = int(rng.choice([1, 4, 8, 20, 64, 100]))
        budget = int(rng.choice([30, 120, 300]))
        sizes = []
        while sum(sizes) < budget:
            sizes.append(int(rng.integers(1, amax + 1)))
        To, Tp = int(rng.choice([2, 5, 8])), int(rng.choice([1, 3, 8, 12]))
        H = int(rng.choice([64, 64, 32, 128, 80]))
        social, U = bool(rng.random() < 0.85), int(rng.choice([0, 1, 2]))
        s_t, s_w = int(rng.integers(1 << 30)), int(rng.integers(1 << 30))
        zv, ov = float(rng.uniform(0, 0.1)), float(rng.uniform(0.9, 1.0))
        cfg = dict(sizes="%d scenes, %d agents, max %d" % (len(sizes), sum(sizes), max(sizes)), To=To, Tp=Tp, H=H, social=social, U=U)
        t0 = time.perf_counter()
        try:
            t = sw.synth_tracks(len(sizes) + 2, sizes + [2, 2], To, Tp, seed=s_t)
            data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
            B, sb = int(np.sum(sizes)), data.the_batches[:len(sizes)]
            torch.manual_seed(s_w)
            if H <= 64:
                G = sw.Generator(H, 1, use_social=social, device="cuda:0")
                D = sw.Discriminator(Tp, H, 2, device="cuda:0")
            else:
                from socialways_amd import generic
                G = generic.Generator(H, 1, use_social=social).to("cuda:0")
                D = generic.Discriminator(Tp, H, 2).to("cuda:0")
            og = torch.optim.Adam(G.predictor_params(), lr=1e-4, betas=(0.9, 0.999))
            od = torch.optim.Adam(D.parameters(), lr=1e-3, betas=(0.9, 0.999))
            orc = O.SocialWaysOracle(Tp, hidden_size=H, use_social=social, n_unrolling_steps=U)
            orc.load_state(dict(attentioner_dict=G.attention.state_dict(), feature_embedder_dict=G.feature_embedder.state_dict(),
                                encoder_dict=G.encoder.state_dict(), decoder_dict=G.decoder.state_dict(), D_dict=D.state_dict()))
            for k in ("attention", "feature_embedder", "encoder", "decoder", "D"):
                getattr(orc, k).load_state_dict({a: b.cpu() for a, b in getattr(orc, k).state_dict().items()})
            noise = torch.rand(B, H // 2)
            zc = noise.cuda()
            mse = torch.nn.MSELoss()
            obsv, pred = data.obsv[:B], data.pred[:B]
            o4, p4 = sw.get_traj_4d(obsv, pred)
            zeros, ones = torch.zeros(B, 1, device="cuda") + zv, torch.ones(B, 1, device="cuda") * ov
            lb, backup = [], None
            for u in range(U + 1):
                D.zero_grad()
                with torch.no_grad():
                    ph = G(obsv, zc, Tp, sb)
                fl, code = D(o4, ph)
                d_fake, d_info = mse(fl, zeros), mse(code.squeeze(), zc[:, :2])
                rl, _ = D(o4, p4)
                d_real = mse(rl, ones)
                (d_fake + d_real + 0.5 * d_info).backward()
                od.step()
                lb += [d_fake.item(), d_info.item(), d_real.item()]
                if u == 0 and U > 0:
                    backup = copy.deepcopy(D)
            D.zero_grad(); og.zero_grad()
            ph = G(obsv, zc, Tp, sb)
            gl, code = D(o4, ph)
            g_l2, g_fool, g_info = mse(ph[:, :, :2], pred), mse(gl, ones), mse(code.squeeze(), zc[:, :2])
            (g_fool + 0.5 * g_info).backward()
            og.step()
            if backup is not None:
                D.load(backup)
            lb += [g_l2.item(), g_fool.item(), g_info.item()]
            rec = {}
            want, ade, fde = orc.train_step(obsv.cpu(), pred.cpu(), sb, zv, ov, noise, data.ss, record=rec)
            e_l = float(np.max(np.abs(np.asarray(lb) - np.asarray(want)) / (np.abs(np.asarray(want)) + 1e-6)))
            e_g, perp = 0.0, []
            for name in ("attention", "feature_embedder", "encoder", "decoder"):
                mod = getattr(G, name)
                sdg = {k: p for k, p in mod.named_parameters()}
                for k, w in ((kk[len(name) + 1:], vv) for kk, vv in rec["g_grads"].items() if kk.startswith(name + ".")):
                    p = sdg.get(k)
                    if p is None or p.grad is None or p.grad.shape != w.shape:
                        continue
                    e1 = float((p.grad.cpu() - w).abs().max()) / max(float(w.abs().max()), 1e-12)
                    perp.append("    %-34s |g|max %.3e err/max %.2e" % (name + "." + k, float(w.abs().max()), e1))
                    e_g = max(e_g, e1)
            e_w = 0.0
            for name, mod, lr in (("attention", G.attention, 1e-4), ("feature_embedder", G.feature_embedder, 1e-4),
                                  ("encoder", G.encoder, 1e-4), ("decoder", G.decoder, 1e-4), ("D", D, 1e-3)):
                ref = getattr(orc, name).state_dict()
                for k, v in mod.state_dict().items():
                    e_w = max(e_w, float((v.cpu() - ref[k]).abs().max()) / lr)       # in units of one Adam step
            ok = e_l < 1e-4 and e_g < 5e-3 and e_w < 2.2 * (U + 1)
            print("%s #%02d %s | losses %.1e  dG %.1e  weights %.2f lr | %.1fs" % ("ok  " if ok else "FAIL", it, cfg, e_l, e_g, e_w,
                                                                                   time.perf_counter() - t0), flush=True)
            fails += 0 if ok else 1
            if not ok:
                print("\n".join(perp), flush=True)
        except Exception as e:
            if isinstance(e, sw.SocialWaysHipError) and "not supported" in str(e):
                print("skip #%02d %s: %s" % (it, cfg, str(e)[:90]), flush=True)
                continue
            fails += 1
            print("EXC  #%02d %s\n%s" % (it, cfg, traceback.format_exc()), flush=True)
    print("%d configurations, %d failures" % (N, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
