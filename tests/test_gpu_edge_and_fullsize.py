"""Edge cases and full-size (BASELINE metric shape) properties of the HIP path.
Small odd shapes are compared with the CPU oracle directly (no golden needed: the oracle is pinned
to the reference by tests/test_oracle_golden.py); the full 256x8 shape is checked against the oracle
once (forward + one training step) and through size-independent properties (determinism, scene
locality, sub-batch equivalence)."""
import numpy as np
import pytest
import torch

import sw_oracle as O
from _util import assert_close

pytestmark = pytest.mark.gpu


def pair(n_next, use_social=True, seed=0, **kw):
    import socialways_amd as sw
    torch.manual_seed(seed)
    tr = sw.SocialWaysTrainer(n_next, use_social=use_social, device="cuda:0", **kw)
    orc = O.SocialWaysOracle(n_next, use_social=use_social)
    orc.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
    return tr, orc


@pytest.mark.parametrize("sizes,To,Tp", [([1, 5, 16, 2, 13], 3, 5), ([7], 2, 1), ([1, 1, 1], 8, 12), ([64, 3], 8, 12),
                                         ([2] * 9 + [3], 5, 7), ([4, 6, 9], 4, 20)])
def test_step_matches_oracle_on_odd_shapes(sizes, To, Tp):
    """B not a multiple of the 16-agent tile, single-agent scenes only, a 64-agent scene, To/Tp that are
    not multiples of anything, Tp = 1, a long horizon (Tp = 20: the wide pred_encoder staging path): losses, rollout and ADE/FDE sums of one full step vs the oracle."""
    import socialways_amd as sw
    t = sw.synth_tracks(len(sizes) + 2, sizes + [2, 2], To, Tp, seed=11)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    tr, orc = pair(Tp)
    B = int(np.sum(sizes))
    sb = data.the_batches[:len(sizes)]
    torch.manual_seed(3)
    noise = torch.rand(B, 32)
    out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.07, 0.91, noise, data.ss)
    got = tr.losses_from(out, [B], Tp, data.ss)[0]
    want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.07, 0.91, noise, data.ss)
    assert_close(got, np.asarray(want), 5e-5, 2e-6, "9 MSE terms")
    o = out.double().cpu().numpy()
    assert abs(o[-1, 0] - ade) < 1e-4 * max(1.0, abs(ade)) and abs(o[-1, 1] - fde) < 1e-4 * max(1.0, abs(fde))
    for name, mod in (("encoder", tr.G.encoder), ("decoder", tr.G.decoder), ("D", tr.D)):
        ref = getattr(orc, name).state_dict()
        for k, v in mod.state_dict().items():      # after Adam: elementwise agreement bounded by ~lr (see check_weights)
            bad = (v.cpu() - ref[k]).abs() > 2e-3 * 1.01
            assert bad.float().mean().item() == 0.0, (name, k)


@pytest.mark.parametrize("sizes", [[65], [70, 5, 130, 64, 1], [200, 3]])
def test_scenes_larger_than_64_agents_forward(sizes):
    """Scenes above 64 agents run the row-block kernels (online softmax over the scene): same rollout as the
    oracle, mixed freely with small scenes; deterministic."""
    import socialways_amd as sw
    t = sw.synth_tracks(len(sizes), sizes, seed=5)
    tr, orc = pair(12)
    B = int(np.sum(sizes))
    obsv = torch.from_numpy(t["obsvs"]).cuda()
    sb = np.asarray(t["batches"])
    torch.manual_seed(7)
    z = torch.rand(B, 32)
    with torch.no_grad():
        a = tr.G(obsv, z.cuda(), 12, sb)
        b = tr.G(obsv, z.cuda(), 12, sb)
        ref = orc.predict(obsv.cpu(), z, 12, sb)
    assert torch.equal(a, b)
    assert_close(a.cpu(), ref, 3e-5, 3e-6, "rollout with scenes > 64 agents")


@pytest.mark.parametrize("sizes", [[65], [70, 5, 130, 64, 1], [200, 3]])
def test_scenes_larger_than_64_agents_backward(sizes):
    """Gradients of every generator parameter through the row-block social kernels (scenes above 64 agents,
    mixed with small ones) against the oracle's autograd, for a random cotangent on the rollout."""
    import socialways_amd as sw
    t = sw.synth_tracks(len(sizes), sizes, seed=6)
    tr, orc = pair(12)
    B = int(np.sum(sizes))
    obsv = torch.from_numpy(t["obsvs"]).cuda()
    sb = np.asarray(t["batches"])
    torch.manual_seed(8)
    z, cot = torch.rand(B, 32), torch.randn(B, 12, 4) * 0.1
    out = tr.G(obsv, z.cuda(), 12, sb)
    out.backward(cot.cuda())
    ref = orc.predict(obsv.cpu(), z, 12, sb)
    ref.backward(cot)
    assert_close(out.detach().cpu(), ref.detach(), 3e-5, 3e-6, "rollout")
    for name in ("attention", "feature_embedder", "encoder", "decoder"):
        for (k, p), (_, q) in zip(getattr(tr.G, name).named_parameters(), getattr(orc, name).named_parameters()):
            want = q.grad if q.grad is not None else torch.zeros_like(q)
            # sums over up to 40 000 pairs in fp32 on both sides: tolerance relative to the tensor's largest entry
            assert_close(p.grad.cpu(), want, 3e-4, (3e-4 if B > 150 else 3e-5) * max(float(want.abs().max()), 1e-12), "%s.%s" % (name, k))


@pytest.mark.parametrize("sizes", [[17, 33, 20, 48, 64, 16], [31, 2, 47, 63, 1], [15, 9, 2, 1, 12, 3], [16, 16, 32]])
def test_social_block_without_the_last_embedder_layer_per_pair(sizes):
    """The kernels never form f_ij = fc.4(h2_ij) (sigma_ij = <h2_ij, W3^T Wh_j> + <b3, Wh_j>, dW3 / db3 /
    dWh through Q_j = sum_i dsigma_ij h2_ij).  Rollout and every generator gradient against the oracle's autograd (which does
    form f_ij) on scene sizes on both sides of the 16-agent blocks: ragged dense scenes (in-register weight gradients) and
    small scenes (pair rows + per-agent rows for the deferred GEMM), single-agent scenes in between."""
    import socialways_amd as sw
    t = sw.synth_tracks(len(sizes), sizes, seed=15)
    tr, orc = pair(12)
    B = int(np.sum(sizes))
    obsv = torch.from_numpy(t["obsvs"]).cuda()
    sb = np.asarray(t["batches"])
    torch.manual_seed(18)
    z, cot = torch.rand(B, 32), torch.randn(B, 12, 4) * 0.1
    out = tr.G(obsv, z.cuda(), 12, sb)
    out.backward(cot.cuda())
    ref = orc.predict(obsv.cpu(), z, 12, sb)
    ref.backward(cot)
    assert_close(out.detach().cpu(), ref.detach(), 3e-5, 3e-6, "rollout")
    for name in ("attention", "feature_embedder", "encoder", "decoder"):
        for (k, p), (_, q) in zip(getattr(tr.G, name).named_parameters(), getattr(orc, name).named_parameters()):
            want = q.grad if q.grad is not None else torch.zeros_like(q)
            assert_close(p.grad.cpu(), want, 3e-4, 3e-5 * max(float(want.abs().max()), 1e-12), "%s.%s" % (name, k))


def test_training_step_with_large_scenes_matches_oracle():
    """One whole GAN step on a packed batch that holds 100- and 70-agent scenes next to small ones."""
    import socialways_amd as sw
    sizes = [100, 8, 70, 2, 1]
    t = sw.synth_tracks(len(sizes) + 2, sizes + [2, 2], 8, 12, seed=12)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    tr, orc = pair(12)
    B = int(np.sum(sizes))
    sb = data.the_batches[:len(sizes)]
    torch.manual_seed(3)
    noise = torch.rand(B, 32)
    out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.07, 0.91, noise, data.ss)
    got = tr.losses_from(out, [B], 12, data.ss)[0]
    want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.07, 0.91, noise, data.ss)
    assert_close(got, np.asarray(want), 5e-5, 2e-6, "9 MSE terms")
    o = out.double().cpu().numpy()
    assert abs(o[-1, 0] - ade) < 1e-4 * max(1.0, abs(ade)) and abs(o[-1, 1] - fde) < 1e-4 * max(1.0, abs(fde))
    for name, mod in (("encoder", tr.G.encoder), ("decoder", tr.G.decoder), ("feature_embedder", tr.G.feature_embedder),
                      ("attention", tr.G.attention), ("D", tr.D)):
        ref = getattr(orc, name).state_dict()
        for k, v in mod.state_dict().items():      # after Adam: elementwise agreement bounded by ~lr
            assert ((v.cpu() - ref[k]).abs() > 2e-3 * 1.01).float().mean().item() == 0.0, (name, k)


def test_empty_sub_batches_means_one_scene():
    """predict(..., sub_batches=[]) treats the whole batch as one scene (train.py:405-406)."""
    import socialways_amd as sw
    t = sw.synth_tracks(1, 9, seed=2)
    G = sw.Generator(use_social=True, device="cuda:0")
    obsv, z = torch.from_numpy(t["obsvs"]).cuda(), torch.rand(9, 32).cuda()
    with torch.no_grad():
        assert torch.equal(G(obsv, z, 12), G(obsv, z, 12, [[0, 9]]))


def _m1():
    import socialways_amd as sw
    S, A = 256, 8
    t = sw.synth_tracks(S + 64, A, seed=1234)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    sb = data.the_batches[:S]
    return data, sb, S * A


def test_full_size_forward_matches_oracle_and_is_deterministic():
    data, sb, B = _m1()
    tr, orc = pair(12)
    torch.manual_seed(1)
    noise = torch.rand(B, 32)
    with torch.no_grad():
        a = tr.G(data.obsv[:B], noise.cuda(), 12, sb)
        b = tr.G(data.obsv[:B], noise.cuda(), 12, sb)
        ref = orc.predict(data.obsv[:B].cpu(), noise, 12, sb)
    assert torch.equal(a, b), "same inputs -> bitwise identical rollout"
    assert_close(a.cpu(), ref, 3e-5, 3e-6, "M1 rollout vs oracle")


def test_full_size_scene_locality_and_sub_batch_equivalence():
    """The social block is block-diagonal: (a) perturbing scene 0 leaves every other scene's rollout
    bitwise unchanged, (b) running scenes 64..128 alone reproduces their rows of the full batch."""
    data, sb, B = _m1()
    tr, _ = pair(12)
    torch.manual_seed(2)
    z = torch.rand(B, 32).cuda()
    obsv = data.obsv[:B].clone()
    with torch.no_grad():
        full = tr.G(obsv, z, 12, sb)
        obsv2 = obsv.clone()
        obsv2[:8] += 0.05
        pert = tr.G(obsv2, z, 12, sb)
        assert not torch.equal(full[:8], pert[:8]) and torch.equal(full[8:], pert[8:])
        lo, hi = int(sb[64, 0]), int(sb[127, 1])
        part = tr.G(obsv[lo:hi], z[lo:hi], 12, sb[64:128] - lo)
        assert torch.equal(part, full[lo:hi])


def test_full_size_training_step_matches_oracle():
    """One whole GAN step at the BASELINE metric shape (2048 agents, 16384 pairs) vs the oracle."""
    data, sb, B = _m1()
    tr, orc = pair(12)
    torch.manual_seed(4)
    noise = torch.rand(B, 32)
    out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.02, 0.96, noise, data.ss)
    got = tr.losses_from(out, [B], 12, data.ss)[0]
    want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.02, 0.96, noise, data.ss)
    assert_close(got, np.asarray(want), 5e-5, 2e-6, "9 MSE terms at M1")
    o = out.double().cpu().numpy()
    assert abs(o[-1, 0] - ade) / ade < 1e-5 and abs(o[-1, 1] - fde) / fde < 1e-5


def test_graph_replay_equals_eager():
    """The hipGraph-replayed step (captured after two eager steps of a layout) continues the exact same
    trajectory as a purely eager trainer."""
    import socialways_amd as sw
    t = sw.synth_tracks(24, 8, seed=9)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    B = data.n_train_samples
    sb = data.train_batches
    res = []
    for use_graph in (True, False):
        torch.manual_seed(0)
        tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", use_graph=use_graph)
        gen = torch.Generator().manual_seed(5)
        outs = []
        for i in range(6):
            noise = torch.rand(B, 32, generator=gen)
            outs.append(tr.step(data.obsv[:B], data.pred[:B], sb, 0.01 * i, 0.9 + 0.01 * i, noise, data.ss).clone())
        res.append((torch.stack(outs).cpu(), tr.D._flat.clone().cpu(), tr.G._flat_all.clone().cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_step_many_equals_consecutive_steps():
    """K steps in one graph launch (step_many) continue the exact trajectory of K single-step launches."""
    import socialways_amd as sw
    t = sw.synth_tracks(24, 8, seed=9)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    B, sb = data.n_train_samples, data.train_batches
    res = []
    for many in (True, False):
        torch.manual_seed(0)
        tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0")
        gen = torch.Generator().manual_seed(5)
        outs = []
        for rep in range(5):       # eager, eager, capture, replay, replay (x3 steps each)
            batch = [(data.obsv[:B], data.pred[:B], 0.01 * (3 * rep + j), 0.9 + 0.01 * j, torch.rand(B, 32, generator=gen))
                     for j in range(3)]
            if many:
                outs += [o.clone() for o in tr.step_many(batch, sb, data.ss)]
            else:
                outs += [tr.step(o, p, sb, zv, ov, nz, data.ss).clone() for o, p, zv, ov, nz in batch]
        res.append((torch.stack(outs).cpu(), tr.D._flat.clone().cpu(), tr.G._flat_all.clone().cpu(),
                    tr.D_optimizer.t, tr.predictor_optimizer.t))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert res[0][3] == res[1][3] == 30 and res[0][4] == res[1][4] == 15


# ---- BASELINE configs 2 and 4 at full size --------------------------------------------------------------------------
def _step_vs_oracle(S, A, seed, grads_rtol, frac_ok=0.999):
    """One whole GAN step (2 D updates + 1 G update) of the HIP path vs the block-diagonal oracle on the same weights,
    noise and label scalars: the 9 MSE terms, the ADE/FDE sums, the gradients the last D update and the G update saw
    (relative to each tensor's largest entry: fp32 sums over up to 2.1 M pairs on both sides) and the weights after
    Adam (elementwise within ~lr, see tests/test_gpu_trainer.py::check_weights)."""
    import socialways_amd as sw
    if np.isscalar(A):
        t = sw.synth_tracks(S + 2, A, 8, 12, seed=seed)
        B = S * A
    else:                      # ragged: per-scene agent counts
        assert len(A) == S
        t = sw.synth_tracks(S, list(A), 8, 12, seed=seed)
        B = int(np.sum(A))
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    tr, orc = pair(12)
    sb = data.the_batches[:S]
    torch.manual_seed(4)
    noise = torch.rand(B, 32)
    rec = {}
    out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.02, 0.96, noise, data.ss)
    got = tr.losses_from(out, [B], 12, data.ss)[0]
    want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.02, 0.96, noise, data.ss, record=rec)
    assert_close(got, np.asarray(want), 5e-5, 2e-6, "9 MSE terms")
    o = out.double().cpu().numpy()
    assert abs(o[-1, 0] - ade) / ade < 1e-5 and abs(o[-1, 1] - fde) / fde < 1e-5
    assert_close(tr.last_pred_hat.cpu(), rec["pred_hat_4d"], 3e-5, 3e-6, "rollout")
    for name in ("attention", "feature_embedder", "encoder", "decoder"):
        for k, p in getattr(tr.G, name).named_parameters():
            w = rec["g_grads"][name + "." + k]
            assert_close(p.grad.cpu(), w, grads_rtol, grads_rtol * max(float(w.abs().max()), 1e-12), "dG %s.%s" % (name, k))
    # The LAST D update's gradients are taken at weights that already went through one Adam step, whose first update moves
    # every weight by ~lr * sign(g): where a noise-level gradient has the other sign the two sides' weights differ by
    # 2e-3.  So they are compared at IDENTICAL weights: the HIP step keeps deepcopy(D) after the first update
    # (train.py:498-499: the workspace "d_backup" - exactly the weights its second pass ran with); the oracle's D is
    # given those weights and differentiates the same d_loss (train.py:482-495) on its own rollout.
    nD = tr.D._flat.numel()
    Dh = sw.Discriminator(12, 64, 2, device="cuda:0")
    Dh._flat.copy_(tr.ws.buf["d_backup"][:nD])
    Do = O.Discriminator(12, 64, 2)
    Do.load_state_dict({k: v.cpu() for k, v in Dh.state_dict().items()})
    mse = torch.nn.MSELoss()
    o4, p4 = O.get_traj_4d(data.obsv[:B].cpu(), data.pred[:B].cpu())
    fake, code = Do(o4, tr.last_pred_hat.cpu())
    real, _ = Do(o4, p4)
    (mse(fake, torch.zeros(B, 1) + 0.02) + mse(real, torch.ones(B, 1) * 0.96) + 0.5 * mse(code.squeeze(), noise[:, :2])).backward()
    for (k, p), (_, q) in zip(tr.D.named_parameters(), Do.named_parameters()):
        w = q.grad
        assert_close(p.grad.cpu(), w, grads_rtol, grads_rtol * max(float(w.abs().max()), 1e-12), "dD (update 2, same weights) %s" % k)
    # Weights after the step.  Adam's first updates move a weight by lr * sign(g) whatever |g| is, so an element whose
    # gradient is at rounding level may land 2 lr (4 lr after D's two updates) away from the oracle's; everything else
    # must agree to a small fraction of one update.  Hard bound per module: 2.2 updates' worth; and all but a small
    # fraction of every tensor within 5 % of lr.
    lr_of = {"encoder": 1e-4, "decoder": 1e-4, "feature_embedder": 1e-4, "attention": 1e-4, "D": 1e-3}
    for name, mod in (("encoder", tr.G.encoder), ("decoder", tr.G.decoder), ("feature_embedder", tr.G.feature_embedder),
                      ("attention", tr.G.attention), ("D", tr.D)):
        ref = getattr(orc, name).state_dict()
        lr, n_upd = lr_of[name], (2 if name == "D" else 1)
        for k, v in mod.state_dict().items():
            d = (v.cpu() - ref[k]).abs()
            assert float(d.max()) <= 2.2 * lr * n_upd, (name, k, float(d.max()))
            close = float((d <= 0.05 * lr + 1e-6 * ref[k].abs()).float().mean())
            assert close >= frac_ok, "%s.%s: only %.4f of the elements within 5 %% of lr" % (name, k, close)
    return tr, data, sb, B


def test_m1_training_step_gradients_and_weights_match_oracle():
    """The BASELINE metric shape itself (256 scenes x 8 agents = 2 048 agents: exactly 128 agent tiles, the one-launch
    discriminator update, weight-gradient jobs sized for <= 256 CUs): gradients at identical weights and the weights after
    the step, as for c2 / c4."""
    _step_vs_oracle(256, 8, 11, 2e-4)


def test_c3_ragged_real_shaped_step_matches_oracle():
    """A packed batch shaped like the real recordings: ~455 scenes of 1..8 agents (single-agent scenes included), 2 048
    agents - tiles that straddle scenes, half-empty social tiles, scenes that own no pairs."""
    import socialways_amd as sw
    sizes = sw.ragged_scene_sizes(2048, 8, seed=77)
    assert min(sizes) == 1 and max(sizes) == 8 and sum(sizes) == 2048
    _step_vs_oracle(len(sizes), sizes, 13, 2e-4)


def test_c2_training_step_matches_oracle():
    """BASELINE config 2 (`--batch-size 256`: 32 scenes x 8 agents): 16 agent tiles on 256 CUs."""
    _step_vs_oracle(32, 8, 21, 2e-4)


def test_c4_full_size_training_step_matches_oracle():
    """BASELINE config 4 at FULL size - 512 scenes x 64 agents = 32 768 agents, 2.1 M pairs in one packed batch
    (the reference itself cannot run it; the block-diagonal oracle does, ~10 s of host time).  This is
    the path of 2 048 agent tiles, 1 024-workgroup weight-gradient rounds, in-register pair-MLP gradients and the
    multi-block second-stage reductions."""
    tr, data, sb, B = _step_vs_oracle(512, 64, 31, 2e-3)      # gradient tolerance: see the kink note in the next test
    # the same step again from the same state is bitwise reproducible (fixed-order reductions everywhere)
    st = {k: (v if not isinstance(v, dict) else {a: (b.clone() if torch.is_tensor(b) else b) for a, b in v.items()})
          for k, v in tr.checkpoint().items()}
    torch.manual_seed(9)
    noise = torch.rand(B, 32)
    outs = []
    for rep in range(2):
        tr.load_checkpoint(st)
        outs.append((tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.95, noise, data.ss, out=False).clone(),
                     tr.G._flat_all.clone(), tr.D._flat.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))


def test_c4_full_size_forward_locality_and_sub_batch_equivalence():
    """Dense crowd at full size: rollout vs the oracle, scene locality (perturbing scene 0 changes nothing else) and
    sub-batch equivalence (scenes 100..131 alone reproduce their rows of the 512-scene batch) - bit for bit."""
    import socialways_amd as sw
    S, A = 512, 64
    t = sw.synth_tracks(S, A, 8, 12, seed=32)
    tr, orc = pair(12)
    B = S * A
    obsv = torch.from_numpy(t["obsvs"]).cuda()
    sb = np.asarray(t["batches"])
    torch.manual_seed(2)
    z = torch.rand(B, 32)
    with torch.no_grad():
        full = tr.G(obsv, z.cuda(), 12, sb)
        ref = orc.predict(obsv.cpu(), z, 12, sb)
        assert_close(full.cpu(), ref, 3e-5, 3e-6, "c4 rollout vs oracle")
        obsv2 = obsv.clone()
        obsv2[:A] += 0.05
        pert = tr.G(obsv2, z.cuda(), 12, sb)
        assert not torch.equal(full[:A], pert[:A]) and torch.equal(full[A:], pert[A:])
        lo, hi = int(sb[100, 0]), int(sb[131, 1])
        part = tr.G(obsv[lo:hi], z[lo:hi].cuda(), 12, sb[100:132] - lo)
        assert torch.equal(part, full[lo:hi])
    # same-weight gradients at full size: generator (random cotangent on the rollout) and discriminator (random
    # cotangents on label / code).  A gradient entry here is a sum over up to 2.1 M pair rows of terms of both signs, so
    # the reference value is the oracle evaluated in FLOAT64.  Tolerance 2e-3 of each tensor's largest entry: a batch
    # of this size evaluates ~1e8 LeakyReLU / ReLU units, and a handful of their pre-activations land within fp32
    # rounding of the kink (measured on this very input: agent 410 has a decoder pre-activation of 7e-9), where any
    # two fp32 evaluation orders disagree on the sign - the slope through that unit then differs by a factor 5 for
    # one agent-step, ~1 % of that agent's gradient and a few 1e-4 of the batch gradient.  Away from such inputs
    # the HIP gradients are within 1e-6 of the float64 oracle (tools/dbg/grad_err.py prints both).
    cot = torch.randn(B, 12, 4) * 0.1
    out = tr.G(obsv, z.cuda(), 12, sb)
    out.backward(cot.cuda())
    torch.set_default_dtype(torch.float64)
    try:
        for m in (orc.attention, orc.feature_embedder, orc.encoder, orc.decoder):
            m.double()
        ref = orc.predict(obsv.cpu().double(), z.double(), 12, sb)
        ref.backward(cot.double())
    finally:
        torch.set_default_dtype(torch.float32)
    for name in ("attention", "feature_embedder", "encoder", "decoder"):
        for (k, p), (_, q) in zip(getattr(tr.G, name).named_parameters(), getattr(orc, name).named_parameters()):
            want = q.grad if q.grad is not None else torch.zeros_like(q)
            assert_close(p.grad.cpu(), want, 2e-3, 2e-3 * max(float(want.abs().max()), 1e-12), "dG %s.%s" % (name, k))
    o4, p4 = sw.get_traj_4d(obsv, torch.from_numpy(t["preds"]).cuda())
    cl, cc = torch.randn(B, 1), torch.randn(B, 2)
    lab, code = tr.D(o4, p4)
    (lab * cl.cuda()).sum().add((code * cc.cuda()).sum()).backward()
    lab_r, code_r = orc.D(o4.cpu(), p4.cpu())
    (lab_r * cl).sum().add((code_r * cc).sum()).backward()
    assert_close(lab.detach().cpu(), lab_r.detach(), 3e-5, 3e-6, "D label")
    for (k, p), (_, q) in zip(tr.D.named_parameters(), orc.D.named_parameters()):
        assert_close(p.grad.cpu(), q.grad, 2e-3, 2e-3 * max(float(q.grad.abs().max()), 1e-12), "dD %s" % k)


# ---- workspace growth under captured graphs -------------------------------------------------------------
def test_graphs_survive_workspace_growth():
    """Layout A is captured, a LARGER layout B then outgrows the shared workspaces, A is replayed again: the
    trajectory must equal a purely eager trainer's (the outgrown buffers are retired, not freed, and the graphs are
    re-captured on the new ones)."""
    import socialways_amd as sw
    ta, tb = sw.synth_tracks(12, 8, seed=9), sw.synth_tracks(40, 8, seed=10)
    da = sw.SceneDataset(ta["obsvs"], ta["preds"], ta["batches"], device="cuda:0")
    db = sw.SceneDataset(tb["obsvs"], tb["preds"], tb["batches"], device="cuda:0")
    res = []
    for use_graph in (True, False):
        torch.manual_seed(0)
        tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", use_graph=use_graph)
        gen = torch.Generator().manual_seed(5)
        outs = []
        for d, n in ((da, 4), (db, 4), (da, 3), (db, 2)):
            B, sbx = d.n_train_samples, d.train_batches
            for i in range(n):
                noise = torch.rand(B, 32, generator=gen)
                outs.append(tr.step(d.obsv[:B], d.pred[:B], sbx, 0.01 * i, 0.9 + 0.01 * i, noise, d.ss).clone().cpu())
        res.append((outs, tr.D._flat.clone().cpu(), tr.G._flat_all.clone().cpu()))
        if use_graph:
            assert tr._graphs and not tr.ws.retired
    assert all(torch.equal(a, b) for a, b in zip(res[0][0], res[1][0]))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("H", [32, 16, 48])
def test_smaller_hidden_sizes_run_zero_padded_and_match_the_oracle(H, tmp_path):
    """`--hidden-size` below the default 64 (train.py:42-44, 76-81): the network runs on the same kernels with its
    weights zero-padded to 64 hidden units - exactly, a padded unit never leaves 0.  Same initial weights as the
    reference would draw (construction order and RNG stream), the checkpoint in the reference's shapes, two whole GAN
    steps against the oracle built with that hidden size (9 MSE terms, rollout, ADE/FDE, the gradients of the second
    step in the reference's shapes), the padding still exactly zero afterwards, and a resume from the saved file."""
    import socialways_amd as sw
    torch.manual_seed(7)
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0")
    torch.manual_seed(7)
    orc = O.SocialWaysOracle(12, hidden_size=H, use_social=True)
    ck = tr.checkpoint()
    for name, mod in (("encoder_dict", orc.encoder), ("decoder_dict", orc.decoder), ("feature_embedder_dict", orc.feature_embedder),
                      ("attentioner_dict", orc.attention), ("D_dict", orc.D)):
        for k, v in mod.state_dict().items():      # identical initialisation, reference shapes
            assert tuple(ck[name][k].shape) == tuple(v.shape) and torch.equal(ck[name][k].cpu(), v), (name, k)
    t = sw.synth_tracks(8, [5, 1, 9, 16, 3, 2, 2, 2], 8, 12, seed=5)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    B, sb = 36, data.the_batches[:6]
    gen = torch.Generator().manual_seed(2)
    for it in range(2):
        noise = torch.rand(B, H // 2, generator=gen)
        rec = {}
        out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
        got = tr.losses_from(out, [B], 12, data.ss)[0]
        want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.03, 0.94, noise, data.ss, record=rec)
        assert_close(got, np.asarray(want), 1e-4 if it else 5e-5, 3e-6, "9 MSE terms, step %d" % it)
        o = out.double().cpu().numpy()
        assert abs(o[-1, 0] - ade) / ade < 2e-5 and abs(o[-1, 1] - fde) / fde < 2e-5
        assert_close(tr.last_pred_hat.cpu(), rec["pred_hat_4d"], 1e-4, 1e-5, "rollout, step %d" % it)
        if it == 0:      # same-weight gradients of the first step, in the reference's shapes
            for name in ("attention", "feature_embedder", "encoder", "decoder"):
                mod = getattr(tr.G, name)
                for i, (k, p) in enumerate(mod.named_parameters()):
                    w = rec["g_grads"][name + "." + k]
                    g = mod.true_view(i, p.grad)
                    assert_close(g.cpu(), w, 2e-4, 2e-4 * max(float(w.abs().max()), 1e-12), "dG %s.%s" % (name, k))
    for mod in (tr.G.encoder, tr.G.decoder, tr.G.feature_embedder, tr.G.attention, tr.D):
        assert float((mod._flat * (1 - mod.pad_mask())).abs().max()) == 0.0, "zero padding stayed exactly zero: %s" % type(mod).__name__
    # checkpoint in the reference's shapes (incl. both Adam dicts), resume in a fresh trainer: same next step
    path = tmp_path / "h.pt"
    tr.save(str(path), epoch=3)
    ck = torch.load(str(path))
    for k, v in orc.decoder.state_dict().items():
        assert tuple(ck["decoder_dict"][k].shape) == tuple(v.shape)
    assert tuple(ck["pred_optimizer"]["state"][0]["exp_avg"].shape) == (H, H)          # attention W, reference shape
    assert tuple(ck["pred_optimizer"]["state"][6]["exp_avg_sq"].shape) == (H, 64)      # feature_embedder fc.4.weight
    assert tuple(ck["D_optimizer"]["state"][1]["exp_avg"].shape) == (4 * H, H)         # D's LSTM weight_hh
    tr2 = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0")
    assert tr2.load_checkpoint(str(path)) == 4
    noise = torch.rand(B, H // 2, generator=gen)
    a = tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.9, noise, data.ss)
    b = tr2.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.9, noise, data.ss)
    assert torch.equal(a, b) and torch.equal(tr.G._flat_all, tr2.G._flat_all) and torch.equal(tr.D._flat, tr2.D._flat)


@pytest.mark.parametrize("H,nl", [(128, 2), (96, 2), (64, 3), (80, 2)])      # 80 units (decoder 200 / 100 / 50): generic path
def test_wider_networks_train_on_the_generic_path_and_match_the_oracle(H, nl, tmp_path):
    """`--hidden-size` above the fused kernels' 64 units (train.py:42-44, 76-81) and latent-code counts other than 2
    (train.py:65): SocialWaysTrainer hands these to the wide path (socialways_amd/wide.py: time-step-level kernels, explicit
    backward; these three widths are multiples of 32) - a subclass of the generic-width trainer (socialways_amd/generic.py).  Same initial weights as the reference draws, two whole GAN steps against the
    oracle built with that width (9 MSE terms, rollout, ADE/FDE, the generator's gradients of the first step, the
    weights after both steps), a checkpoint in the reference's format and a resume from the file."""
    import socialways_amd as sw
    from socialways_amd.generic import GenericTrainer
    torch.manual_seed(7)
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, n_latent_codes=nl, device="cuda:0")
    assert isinstance(tr, GenericTrainer)
    torch.manual_seed(7)
    orc = O.SocialWaysOracle(12, hidden_size=H, use_social=True, n_latent_codes=nl)
    ck = tr.checkpoint()
    for name, mod in (("encoder_dict", orc.encoder), ("decoder_dict", orc.decoder), ("feature_embedder_dict", orc.feature_embedder),
                      ("attentioner_dict", orc.attention), ("D_dict", orc.D)):
        for k, v in mod.state_dict().items():      # identical initialisation, reference shapes and keys
            assert tuple(ck[name][k].shape) == tuple(v.shape) and torch.equal(ck[name][k].cpu(), v), (name, k)
    t = sw.synth_tracks(8, [5, 1, 9, 16, 3, 2, 2, 2], 8, 12, seed=5)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    B, sb = 36, data.the_batches[:6]
    gen = torch.Generator().manual_seed(2)
    for it in range(2):
        noise = torch.rand(B, H // 2, generator=gen)
        rec = {}
        out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
        got = tr.losses_from(out, [B], 12, data.ss)[0]
        want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.03, 0.94, noise, data.ss, record=rec)
        assert_close(got, np.asarray(want), 1e-4 if it else 5e-5, 3e-6, "9 MSE terms, step %d" % it)
        o = out.double().cpu().numpy()
        assert abs(o[-1, 0] - ade) / ade < 2e-5 and abs(o[-1, 1] - fde) / fde < 2e-5
        assert_close(tr.last_pred_hat.cpu(), rec["pred_hat_4d"], 1e-4, 1e-5, "rollout, step %d" % it)
        if it == 0:      # same-weight gradients of the first step
            for name in ("attention", "feature_embedder", "encoder", "decoder"):
                for k, p in getattr(tr.G, name).named_parameters():
                    w = rec["g_grads"][name + "." + k]
                    assert_close(p.grad.cpu(), w, 2e-4, 2e-4 * max(float(w.abs().max()), 1e-12), "dG %s.%s" % (name, k))
    for name, mod in (("encoder", tr.G.encoder), ("decoder", tr.G.decoder), ("D", tr.D)):
        ref = getattr(orc, name).state_dict()
        for k, v in mod.state_dict().items():      # two Adam steps: within a fraction of lr except noise-level gradients
            lr = 1e-3 if name == "D" else 1e-4
            d = (v.cpu() - ref[k]).abs()
            assert float(d.max()) <= 4.4 * lr and float((d <= 0.1 * lr).float().mean()) > 0.95, (name, k, float(d.max()))
    path = tmp_path / "wide.pt"
    tr.save(str(path), epoch=3)
    tr2 = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, n_latent_codes=nl, device="cuda:0")
    assert tr2.load_checkpoint(str(path)) == 4
    noise = torch.rand(B, H // 2, generator=gen)
    a = tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.9, noise, data.ss)
    b = tr2.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.9, noise, data.ss)
    assert torch.equal(a, b)
    for p, q in zip(tr.G.parameters(), tr2.G.parameters()):
        assert torch.equal(p, q)
    # evaluation on the generic path: test() of train.py:563-616
    res = tr.test(data, n_gen_samples=3)
    assert all(np.isfinite(res))


def test_wide_path_equals_generic_path_and_graph_replay_equals_eager():
    """The wide engine (wide.py: time-step-level kernels, explicit backward) against the layer-by-layer generic path under
    torch's tape (generic.py) on the same modules: reported sums, rollout, EVERY gradient of the generator, the weights
    of G and D after each of three steps; then its hipGraph-replayed step against its eager step, bit for bit.  Widths
    that are not multiples of 32 stay on the generic path."""
    import socialways_amd as sw
    from socialways_amd.generic import GenericTrainer
    from socialways_amd.wide import WideTrainer
    H = 128
    assert type(sw.SocialWaysTrainer(12, hidden_size=H, device="cuda:0")) is WideTrainer
    assert type(sw.SocialWaysTrainer(12, hidden_size=80, device="cuda:0")) is GenericTrainer
    torch.manual_seed(7)
    a = WideTrainer(12, hidden_size=H, device="cuda:0", use_graph=False)
    torch.manual_seed(7)
    b = GenericTrainer(12, hidden_size=H, device="cuda:0")
    t = sw.synth_tracks(8, [5, 1, 9, 16, 3, 2, 2, 2], 8, 12, seed=5)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    B, sb = 36, data.the_batches[:6]
    gen = torch.Generator().manual_seed(2)
    for it in range(3):
        noise = torch.rand(B, H // 2, generator=gen)
        ra = a.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
        rb = b.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
        assert_close(ra.cpu(), rb.cpu(), 2e-6, 1e-9, "reported sums, step %d" % it)
        assert_close(a.last_pred_hat.cpu(), b.last_pred_hat.cpu(), 1e-5, 1e-6, "rollout, step %d" % it)
        for (n, p), (_, q) in zip(a.G.named_parameters(), b.G.named_parameters()):
            assert_close(p.grad.cpu(), q.grad.cpu(), 1e-4, 1e-4 * max(float(q.grad.abs().max()), 1e-12), "dG %s, step %d" % (n, it))
        for (n, p), (_, q) in zip(list(a.G.named_parameters()) + list(a.D.named_parameters()),
                                  list(b.G.named_parameters()) + list(b.D.named_parameters())):
            lr = 1e-3 if n in dict(a.D.named_parameters()) else 1e-4
            dpq = (p.detach() - q.detach()).abs()
            assert float(dpq.max()) <= 2.2 * lr * (it + 1), n        # Adam: sign flips of noise-level gradients
            assert float((dpq <= 0.1 * lr).float().mean()) > 0.98, n
    torch.manual_seed(7)
    c = WideTrainer(12, hidden_size=H, device="cuda:0", use_graph=True)
    torch.manual_seed(7)
    d = WideTrainer(12, hidden_size=H, device="cuda:0", use_graph=False)
    gen = torch.Generator().manual_seed(3)
    for it in range(6):           # eager, eager, capture + replay, replay x 3
        noise = torch.rand(B, H // 2, generator=gen)
        rc = c.step(data.obsv[:B], data.pred[:B], sb, 0.01 * it, 0.94, noise, data.ss)
        rd = d.step(data.obsv[:B], data.pred[:B], sb, 0.01 * it, 0.94, noise, data.ss)
        assert torch.equal(rc, rd) and torch.equal(c.gp.flat, d.gp.flat) and torch.equal(c.dp.flat, d.dp.flat), it
    assert len(c._graphs) == 1 and c.D_optimizer.t == 12 and c.predictor_optimizer.t == 6


def test_unsupported_module_widths_say_where_to_go():
    import socialways_amd as sw
    with pytest.raises(sw.SocialWaysHipError, match="generic"):
        sw.DecoderFC(128 + 128 + 64)
    with pytest.raises(sw.SocialWaysHipError):
        sw.EncoderLstm(20, 1)


@pytest.mark.parametrize("H,nl", [(64, 2), (128, 1), (32, 3)])
def test_encoder_with_stacked_layers_or_wide_units_matches_the_reference_module(H, nl):
    """EncoderLstm(hidden_size) with the class signature's default of 2 stacked layers (train.py:246), and widths above 64:
    served by the generic-width module - sequence call, single-step calls from the stored state and all gradients against
    the oracle's nn.LSTM-based restatement with the same parameters."""
    import socialways_amd as sw
    from socialways_amd import generic
    torch.manual_seed(11)
    enc = sw.EncoderLstm(H) if nl == 2 else sw.EncoderLstm(H, nl)
    assert isinstance(enc, generic.EncoderLstm) and enc.n_layers == nl
    ref = O.EncoderLstm(H, nl)
    ref.load_state_dict(enc.state_dict())                     # same keys, same shapes
    enc = enc.to("cuda:0")
    bs, T = 37, 6
    x = torch.randn(bs, T, 4)
    h0, c0 = 0.3 * torch.randn(nl, bs, H), 0.3 * torch.randn(nl, bs, H)
    xg = x.cuda().requires_grad_(True)
    enc.init_lstm(h0.cuda(), c0.cuda())
    y = enc(xg)
    y1 = enc(x[:, 0].cuda())                                  # one more step from the stored state, (B,4) input
    xr = x.clone().requires_grad_(True)
    ref.init_lstm(h0, c0)
    yr = ref(xr)
    yr1 = ref(x[:, 0])
    assert_close(y.detach().cpu(), yr.detach(), 2e-5, 2e-6, "y")
    assert_close(y1.detach().cpu(), yr1.detach(), 2e-5, 2e-6, "single step")
    for k in range(2):
        assert_close(enc.lstm_h[k].detach().cpu(), ref.lstm_h[k].detach(), 2e-5, 2e-6, "state %d" % k)
    w = torch.randn(bs, T, H)
    (y * w.cuda()).sum().backward()
    (yr * w).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad, 1e-4, 1e-6, "dx")
    for (k, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert_close(p.grad.cpu(), q.grad, 1e-4, 1e-5 * max(float(q.grad.abs().max()), 1e-9), "d" + k)


@pytest.mark.timeout(900)
def test_random_configurations_match_the_oracle():
    """A fixed-seed slice of the randomised sweep (tools/dbg/fuzz_parity.py: ragged scene sizes up to 90 agents, observation
    / prediction lengths, unrolling depth 0-2, loss switches incl. L2 / variety, hidden sizes 64 / 32 / 128 / 80 with 2-3
    latent codes -> fused, wide and generic trainers): two steps each against the CPU oracle - MSE terms, ADE / FDE sums,
    rollout, every generator gradient."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "dbg", "fuzz_parity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.run(14, 2) == 0


def _tool(name):
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", "dbg", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.timeout(600)
@pytest.mark.parametrize("hidden,steps", [(64, 70), (128, 40)])
def test_graph_replay_equals_eager_over_a_random_layout_sequence(hidden, steps):
    """A fixed-seed run of tools/dbg/fuzz_graph.py: a graph-capturing trainer against the same trainer running eagerly over a
    random sequence of ragged packed-batch layouts (recurring, growing, shrinking, more than the caches hold, K-step
    launches) - sums and every weight bit-identical after every step."""
    assert _tool("fuzz_graph").run(hidden, steps, 1) == 0


@pytest.mark.timeout(900)
def test_random_datasets_epochs_and_evaluation_match_the_oracle():
    """A fixed-seed slice of tools/dbg/fuzz_epoch.py: train_epoch() (the reference's greedy scene packing, own RNG streams)
    for two epochs and test() with K sampled futures on random ragged datasets, fused / wide / generic trainers, against the
    CPU oracle: epoch ADE / FDE, every step's MSE terms, packed-batch sizes, min / avg ADE and FDE."""
    assert _tool("fuzz_epoch").run(3, 2) == 0
