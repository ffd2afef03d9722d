"""Pin the CPU oracle (oracle/sw_oracle.py) to the golden vectors captured from the unmodified
reference by oracle/make_golden.py.  CPU only; this is what makes the oracle a valid checker."""
import numpy as np
import pytest
import torch

import sw_oracle as O
from _util import golden, state_from, as_checkpoint, dataset_from, assert_close, VARIANT_KW, variant_expected_losses

# The golden vectors were produced with 8 MKL threads; another thread count moves last ulps
# (SURVEY.md §8c), so comparisons are toleranced, not bitwise.
RT, AT = 2e-5, 2e-6


def make_oracle(g, n_next, use_social, social):
    o = O.SocialWaysOracle(n_next, use_social=use_social, social=social)
    o.load_state(as_checkpoint(state_from(g, "w0.")))
    return o


def test_toy_dataset_shapes():
    g8, g6 = golden("toy_768_8_3"), golden("toy_768_6_3")
    assert g8["obsvs"].shape == (768, 2, 2) and g8["preds"].shape == (768, 2, 2)
    n8 = np.diff(g8["batches"], axis=1).ravel()
    assert len(n8) == 512 and sorted(np.unique(n8).tolist()) == [1, 2, 3]
    n6 = np.diff(g6["batches"], axis=1).ravel()
    assert len(n6) == 128 and (n6 == 6).all()


@pytest.mark.parametrize("tag,social", [("off", "blockdiag"), ("on", "blockdiag"), ("on", "faithful")])
def test_toy_epoch(tag, social):
    g = golden("toy_b64_" + tag)
    toy = golden("toy_768_8_3")
    data = O.load_and_normalise(toy["obsvs"], toy["preds"], toy["batches"])
    assert abs(data["ss"] - float(g["ss"])) < 1e-12
    o = make_oracle(g, 2, tag == "on", social)
    steps = iter(range(len(g["losses"])))

    def draw(bs):
        s = next(steps)
        return float(g["uniform"][s, 0]), float(g["uniform"][s, 1]), torch.from_numpy(g["noise.%d" % s])
    ade, fde, losses, shapes = o.train_epoch(data, int(g["batch_size"]), draw=draw)
    assert [s[0] for s in shapes] == g["step_agents"].tolist()
    assert_close(np.asarray(losses), g["losses"], 5e-5, 1e-6, "per-step MSE terms")
    assert abs(ade - float(g["ade"])) < 1e-5 and abs(fde - float(g["fde"])) < 1e-5
    w1 = state_from(g, "w1.")
    for mod, sd in w1.items():
        cur = getattr(o, mod).state_dict()
        for k, v in sd.items():
            assert_close(cur[k].numpy(), v.numpy(), 1e-4, 2e-6, "w1.%s.%s" % (mod, k))


def test_toy_epoch_own_rng_stream():
    """Same seeds -> the oracle draws the same label noise / z as the reference (train.py:471-473)."""
    g = golden("toy_b64_on")
    toy = golden("toy_768_8_3")
    data = O.load_and_normalise(toy["obsvs"], toy["preds"], toy["batches"])
    torch.manual_seed(0)
    np.random.seed(0)
    o = O.SocialWaysOracle(2, use_social=True)
    w0 = state_from(g, "w0.")
    for mod, sd in w0.items():                                   # seed -> init mapping (train.py:370-385)
        cur = getattr(o, mod).state_dict()
        for k, v in sd.items():
            assert torch.equal(cur[k], v), (mod, k)
    np.random.seed(0)
    ade, fde, losses, _ = o.train_epoch(data, 64)
    assert_close(np.asarray(losses), g["losses"], 5e-5, 1e-6, "losses, own RNG")
    assert abs(ade - float(g["ade"])) < 1e-5


@pytest.mark.parametrize("case,social", [("syn_s16a8_off", "blockdiag"), ("syn_s16a8_on", "blockdiag"), ("syn_big_on", "blockdiag"),
                                         ("syn_s16a8_on", "faithful"), ("syn_ragged_on", "blockdiag"),
                                         ("syn_ragged_on", "faithful")])
def test_one_step_all_intermediates(case, social):
    g = golden(case)
    ds = dataset_from(g)
    data = O.load_and_normalise(ds["obsvs"], ds["preds"], ds["batches"])
    o = make_oracle(g, 12, bool(g["use_social"]), social)
    B = int(g["step_agents"][0])
    sb = data["the_batches"][:data["train_size"]]
    rec = {}
    losses, ade, fde = o.train_step(data["obsv"][:B], data["pred"][:B], sb, float(g["uniform"][0, 0]),
                                    float(g["uniform"][0, 1]), torch.from_numpy(g["noise.0"]), data["ss"], rec)
    assert_close(np.asarray(losses), g["losses"][0], 2e-5, 1e-6, "losses")
    assert_close(rec["hT"], g["hT"], RT, AT, "hT")
    assert_close(rec["S"], g["S"], RT, AT, "S")
    assert_close(rec["pred_hat_4d"], g["pred_hat_4d"], RT, AT, "pred_hat_4d")
    assert_close(rec["gen_labels"], g["g_fake.label"], RT, AT, "g label")
    assert_close(rec["gen_code_hat"], g["g_fake.code"], RT, AT, "g code")
    assert_close(rec["fake_labels0"], g["d0_fake.label"], RT, AT, "d0 fake label")
    assert_close(rec["real_labels0"], g["d0_real.label"], RT, AT, "d0 real label")
    gscale = np.abs(g["dpred_hat_4d"]).max()
    assert_close(rec["dpred_hat_4d"], g["dpred_hat_4d"], 1e-4, 1e-5 * gscale, "dpred_hat_4d")
    for u in range(2):
        for k, v in rec["d_grads"][u].items():
            ref = g["dgrad%d.%s" % (u, k)]
            assert_close(v, ref, 1e-4, 1e-5 * max(np.abs(ref).max(), 1e-12), "dgrad%d.%s" % (u, k))
    for k, v in rec["g_grads"].items():
        ref = g["ggrad." + k]
        assert_close(v, ref, 2e-4, 2e-5 * max(np.abs(ref).max(), 1e-12), "ggrad." + k)
    w1 = state_from(g, "w1.")
    for mod, sd in w1.items():
        cur = getattr(o, mod).state_dict()
        for k, v in sd.items():
            assert_close(cur[k].numpy(), v.numpy(), 1e-4, 2e-6, "w1.%s.%s" % (mod, k))
    assert abs(ade / data["n_train_samples"] - float(g["ade"])) < 1e-5


@pytest.mark.parametrize("name", sorted(VARIANT_KW))
def test_loss_and_unrolling_switches(name):
    """use_l2_loss / use_variety_loss (as written, train.py:527-536) / n_unrolling_steps 0,2 / info loss
    off: one step of the syn_s16a8_on case with the reference's module globals flipped."""
    g, base = golden("syn_variants"), golden("syn_s16a8_on")
    ds = dataset_from(base)
    data = O.load_and_normalise(ds["obsvs"], ds["preds"], ds["batches"])
    o = O.SocialWaysOracle(12, use_social=True, **VARIANT_KW[name])
    o.load_state(as_checkpoint(state_from(g, "w0.")))
    B = int(base["step_agents"][0])
    sb = data["the_batches"][:data["train_size"]]
    rec = {}
    losses, ade, fde = o.train_step(data["obsv"][:B], data["pred"][:B], sb, float(g["uniform"][0, 0]),
                                    float(g["uniform"][0, 1]), torch.from_numpy(g["noise"]), data["ss"], rec)
    want, extra = variant_expected_losses(g, name)
    assert_close(np.asarray(losses), want, 2e-5, 1e-6, "losses")
    if name == "variety":
        assert len(extra) == 20 and abs(rec["variety"] - extra[-1]) < 1e-6    # only k = 19 enters the loss
    assert len(rec["d_grads"]) == int(g[name + ".n_d_updates"])
    for k, v in rec["d_grads"][-1].items():
        ref = g["%s.dgrad_last.%s" % (name, k)]
        assert_close(v, ref, 1e-4, 1e-5 * max(np.abs(ref).max(), 1e-12), "dgrad_last." + k)
    for k, v in rec["g_grads"].items():
        ref = g["%s.ggrad.%s" % (name, k)]
        assert_close(v, ref, 2e-4, 2e-5 * max(np.abs(ref).max(), 1e-12), "ggrad." + k)
    for k, v in o.D.state_dict().items():
        assert_close(v.numpy(), g["%s.w1.D.%s" % (name, k)], 1e-4, 2e-6, "w1.D." + k)
    want_af = g[name + ".ade_fde"] * data["n_train_samples"]
    assert abs(ade / want_af[0] - 1) < 1e-6 and abs(fde / want_af[1] - 1) < 1e-6


def test_social_ops_dense_and_blockdiag():
    g = golden("social_ops")
    st = state_from(g, "w0.")
    fe, att = O.EmbedSocialFeatures(3, 64), O.AttentionPooling(64, 64)
    fe.load_state_dict(st["feature_embedder"])
    att.load_state_dict(st["attention"])
    obsv, h, sb = torch.from_numpy(g["obsv"]), torch.from_numpy(g["h"]), g["batches"]
    x4 = O.get_traj_4d(obsv, [])
    with torch.no_grad():
        feats = O.SocialFeatures(x4, sb)
        assert_close(feats, g["features"], 1e-5, 1e-6, "dense features")
        emb = fe(feats, sb)
        for s, (a, b) in enumerate(sb):
            assert_close(emb[a:b, a:b], g["emb.%d" % s], 1e-5, 2e-6, "emb block %d" % s)
        assert_close(att(emb, h, sb), g["S"], 1e-5, 2e-6, "faithful S")
        assert_close(O.social_pool_blockdiag(x4[:, -1], h, sb, fe, att), g["S"], 1e-5, 2e-6, "blockdiag S")
        last = x4[:, -1]
        for (i, j), dca, bear in zip(g["pairs"], g["dca_scalar"], g["bearing_scalar"]):
            assert abs(O.DCA(last[i], last[j]).item() - dca) < 2e-6            # scalar spec, train.py:192-205
            assert abs(O.Bearing(last[i], last[j]).item() - bear) < 2e-6
            f = O.pair_features(last[i], last[j])
            assert abs(f[2].item() - dca) < 2e-6 and abs(f[1].item() - bear) < 2e-6
    s_single = [i for i, (a, b) in enumerate(sb) if b - a == 1]
    assert s_single and all((g["S"][sb[i][0]] == 0).all() for i in s_single)   # N==1 -> S=0 (train.py:165)


def test_test_eval_and_prediction_npz():
    g = golden("test_eval")
    ds = dataset_from(g)
    data = O.load_and_normalise(ds["obsvs"], ds["preds"], ds["batches"])
    o = make_oracle(g, 12, True, "blockdiag")
    torch.manual_seed(123)
    coll = []
    metrics = o.test(data, n_gen_samples=4, times=g["ds.times"], collect=coll)
    assert_close(np.asarray(metrics), g["metrics"], 1e-5, 1e-6, "test() metrics")
    files = [str(f) for f in g["npz_files"]]
    assert len(files) == len(coll) == 2
    for f, c in zip(files, coll):
        tag = f[:-4]
        assert tag == "1-%s" % c["timestamp"]                                   # '<epoch>-<t>.npz' (train.py:592)
        for k in ("obsvs", "preds_our", "preds_gtt", "preds_lnr"):
            assert_close(c[k], g["npz.%s.%s" % (tag, k)], 1e-5, 1e-5, k)
        assert c["preds_our"].shape[0] == 4


def test_toy_statistics_1nn_and_emd():
    """calc_statistics.py compute_1nn / compute_wasserstein (incl. the cost-matrix mirroring quirk)."""
    g = golden("toy_stats")
    for i in range(len(g["sigmas"])):
        assert_close(O.compute_1nn(g["real"], g["fake.%d" % i]), g["one_nn.%d" % i], 0, 1e-12, "1nn %d" % i)
        assert abs(O.compute_wasserstein(g["real"], g["fake.%d" % i]) - float(g["emd.%d" % i])) < 1e-7
    assert_close(O.compute_1nn(g["g.real"], g["g.fake"], 3), g["g.one_nn"], 0, 1e-12, "1nn generic")
    assert abs(O.compute_wasserstein(g["g.real"], g["g.fake"], 3) - float(g["g.emd"])) < 1e-7


def test_five_toy_epochs_track_the_reference():
    """Five consecutive epochs of train() on the toy set (50 Adam-updated GAN steps): the oracle fed with the
    reference's recorded draws reproduces its per-epoch ADE/FDE and MSE terms; the tolerance grows with the
    horizon (chaotic amplification of fp32 summation-order differences), it does not drift."""
    g = golden("toy_multi")
    toy = golden("toy_768_8_3")
    data = O.load_and_normalise(toy["obsvs"], toy["preds"], toy["batches"])
    o = make_oracle(g, 2, True, "blockdiag")
    for ep in range(int(g["n_epochs"])):
        steps = iter(range(len(g["losses.%d" % ep])))

        def draw(bs, ep=ep, steps=steps):
            s = next(steps)
            return float(g["uniform.%d" % ep][s, 0]), float(g["uniform.%d" % ep][s, 1]), torch.from_numpy(g["noise.%d.%d" % (ep, s)])
        ade, fde, losses, _ = o.train_epoch(data, int(g["batch_size"]), draw=draw)
        tol = 1e-5 * 10 ** ep
        assert abs(ade - float(g["ade.%d" % ep])) < tol and abs(fde - float(g["fde.%d" % ep])) < tol, (ep, ade, fde)
        assert_close(np.asarray(losses), g["losses.%d" % ep], 1e-4 * 10 ** ep, 1e-6, "MSE terms of epoch %d" % (ep + 1))


def test_oracle_fixed_variety_term_reduces_to_l2_for_one_sample_and_is_a_minimum():
    """The oracle's `use_variety_loss="fixed"` (best-of-K L2, NOT reference behaviour; the as-written branch is pinned
    by the `variety` golden above): with K = 1 it is exactly the L2 term (same generator gradients as
    use_l2_loss=True), and with K > 1 the reported value is the mean of the per-agent minima."""
    import sw_oracle as O
    t = O.synth_dataset(6, [3, 1, 4, 2, 2, 2], 8, 12, seed=5)
    data = O.load_and_normalise(t["obsvs"], t["preds"], t["batches"])
    B, sb = 10, data["the_batches"][:4]
    grads = []
    for kw in (dict(use_l2_loss=True), dict(use_variety_loss="fixed", variety_k=1)):
        torch.manual_seed(0)
        orc = O.SocialWaysOracle(12, use_social=True, **kw)
        torch.manual_seed(1)
        noise, rec = torch.rand(B, 32), {}
        orc.train_step(data["obsv"][:B], data["pred"][:B], sb, 0.05, 0.95, noise, data["ss"], record=rec,
                       variety_noise=torch.zeros(0, 32))
        grads.append(rec["g_grads"])
    for k in grads[0]:
        assert torch.allclose(grads[0][k], grads[1][k], rtol=1e-5, atol=1e-9), k
    torch.manual_seed(0)
    orc = O.SocialWaysOracle(12, use_social=True, use_variety_loss="fixed", variety_k=4)
    torch.manual_seed(1)
    noise, vn, rec = torch.rand(B, 32), torch.rand(3 * B, 32), {}
    orc.train_step(data["obsv"][:B], data["pred"][:B], sb, 0.05, 0.95, noise, data["ss"], record=rec, variety_noise=vn)
    l2 = rec["variety_l2"]
    assert l2.shape == (4, B) and abs(float(l2.min(0)[0].mean()) - rec["variety"]) < 1e-7
    assert (rec["variety_kmin"] == l2.argmin(0)).all() and len(set(rec["variety_kmin"].tolist())) > 1


def test_oracle_resumes_from_the_checkpoint_the_reference_wrote():
    """tests/golden/ref_checkpoint.npz: the unmodified reference trained 50 toy epochs, saved its checkpoint itself
    (train.py:651-663), a second reference process loaded it (train.py:622-634) and trained epoch 51.  The oracle,
    loaded from the same file contents incl. both Adam dicts (generator state present for indices 8..21 only),
    reproduces that epoch."""
    import sw_oracle as O
    from _util import reference_checkpoint
    g = golden("ref_checkpoint")
    toy = golden("toy_768_8_3")
    data = O.load_and_normalise(toy["obsvs"], toy["preds"], toy["batches"])
    ck = reference_checkpoint(g)
    assert sorted(ck["pred_optimizer"]["state"]) == list(range(8, 22)) and len(ck["D_optimizer"]["state"]) == 20
    torch.manual_seed(5)
    orc = O.SocialWaysOracle(2, use_social=False)
    orc.load_state(ck)
    orc.predictor_optimizer.load_state_dict(ck["pred_optimizer"])
    orc.D_optimizer.load_state_dict(ck["D_optimizer"])
    draws = iter([(float(u[0]), float(u[1]), torch.from_numpy(g["resume.noise.%d" % s])) for s, u in enumerate(g["resume.uniform"])])
    ade, fde, losses, _ = orc.train_epoch(data, int(g["batch_size"]), draw=lambda bs: next(draws))
    assert_close(np.asarray(losses), g["resume.losses"], 2e-5, 1e-7, "epoch 51 MSE terms")
    assert abs(ade - float(g["resume.ade"])) < 1e-5 and abs(fde - float(g["resume.fde"])) < 1e-5
