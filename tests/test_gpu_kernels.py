"""GPU parity of every HIP kernel family against the golden vectors of the reference and against
the CPU oracle, through the C ABI (socialways_amd/_lib.py).  fp32; tolerances are stated per check:
the kernels use v_mfma_f32_16x16x4_f32 (an exact fp32 fmaf chain), so differences to MKL are
summation-order only."""
import numpy as np
import pytest
import torch

import sw_oracle as O
from _util import golden, state_from, as_checkpoint, dataset_from, assert_close

pytestmark = pytest.mark.gpu
RT, AT = 2e-5, 2e-6


def _models(g, n_next, use_social):
    import socialways_amd as sw
    dev = torch.device("cuda:0")
    G = sw.Generator(use_social=use_social, device=dev)
    D = sw.Discriminator(n_next, 64, 2, device=dev)
    st = state_from(g, "w0.")
    G.attention.load_state_dict(st["attention"])
    G.feature_embedder.load_state_dict(st["feature_embedder"])
    G.encoder.load_state_dict(st["encoder"])
    G.decoder.load_state_dict(st["decoder"])
    D.load_state_dict(st["D"])
    return G, D, dev


def _step_inputs(g):
    ds = dataset_from(g)
    data = O.load_and_normalise(ds["obsvs"], ds["preds"], ds["batches"])
    B = int(g["step_agents"][0])
    sb = data["the_batches"][:data["train_size"]]
    return data, data["obsv"][:B], data["pred"][:B], sb, torch.from_numpy(g["noise.0"])


def cmp_grads(got, g, prefix, rtol, atol_rel):
    for k, v in got.items():
        ref = g[prefix + k]
        assert_close(v.detach().cpu().numpy(), ref, rtol, atol_rel * max(np.abs(ref).max(), 1e-12), prefix + k)


@pytest.mark.parametrize("case", ["syn_s16a8_off", "syn_s16a8_on", "syn_ragged_on", "syn_big_on"])
def test_predict_forward(case):
    g = golden(case)
    G, D, dev = _models(g, 12, bool(g["use_social"]))
    data, obsv, pred, sb, noise = _step_inputs(g)
    with torch.no_grad():
        out = G(obsv.to(dev), noise.to(dev), 12, sb)
    assert_close(out.cpu(), g["pred_hat_4d"], RT, AT, "pred_hat_4d")


@pytest.mark.parametrize("case", ["syn_s16a8_on", "syn_ragged_on"])
def test_disc_forward(case):
    import socialways_amd as sw
    g = golden(case)
    G, D, dev = _models(g, 12, True)
    data, obsv, pred, sb, noise = _step_inputs(g)
    o4, p4 = sw.get_traj_4d(obsv.to(dev), pred.to(dev))
    assert_close(o4.cpu(), g["obsv_4d"], 0, 1e-7, "obsv_4d")
    assert_close(p4.cpu(), g["pred_4d"], 0, 1e-7, "pred_4d")
    with torch.no_grad():
        lab_r, code_r = D(o4, p4)
        lab_f, code_f = D(o4, torch.from_numpy(g["pred_hat_4d"]).to(dev))
    assert_close(lab_r.cpu(), g["d0_real.label"], RT, AT, "real label")
    assert_close(code_r.cpu(), g["d0_real.code"], RT, AT, "real code")
    assert_close(lab_f.cpu(), g["d0_fake.label"], RT, AT, "fake label")
    assert_close(code_f.cpu(), g["d0_fake.code"], RT, AT, "fake code")


@pytest.mark.parametrize("case", ["syn_s16a8_off", "syn_s16a8_on", "syn_ragged_on", "syn_big_on"])
def test_predict_backward(case):
    """dL/dpred_hat of the reference's G phase pushed through the HIP backward: every generator
    gradient must match the reference's autograd."""
    g = golden(case)
    G, D, dev = _models(g, 12, bool(g["use_social"]))
    data, obsv, pred, sb, noise = _step_inputs(g)
    out = G(obsv.to(dev), noise.to(dev), 12, sb)
    out.backward(torch.from_numpy(g["dpred_hat_4d"]).to(dev))
    got = {}
    for name, mod in (("attention", G.attention), ("feature_embedder", G.feature_embedder),
                      ("encoder", G.encoder), ("decoder", G.decoder)):
        for k, p in mod.named_parameters():
            got[name + "." + k] = p.grad
    cmp_grads(got, g, "ggrad.", 2e-4, 2e-5)


@pytest.mark.parametrize("case", ["syn_s16a8_on", "syn_ragged_on"])
def test_disc_backward_first_update(case):
    """D update u=0 of train.py:476-496 through autograd on the HIP Function: d_loss = fake + real +
    0.5 info; gradients vs the reference's."""
    import socialways_amd as sw
    g = golden(case)
    G, D, dev = _models(g, 12, True)
    data, obsv, pred, sb, noise = _step_inputs(g)
    o4, p4 = sw.get_traj_4d(obsv.to(dev), pred.to(dev))
    B = obsv.shape[0]
    zeros = torch.zeros(B, 1, device=dev) + float(g["uniform"][0, 0])
    ones = torch.ones(B, 1, device=dev) * float(g["uniform"][0, 1])
    z = noise.to(dev)
    mse = torch.nn.MSELoss()
    fake, code = D(o4, torch.from_numpy(g["pred_hat_4d"]).to(dev))
    l_fake, l_info = mse(fake, zeros), mse(code.squeeze(), z[:, :2])
    real, _ = D(o4, p4)
    l_real = mse(real, ones)
    assert_close([l_fake.item(), l_info.item(), l_real.item()], g["losses"][0][:3], 2e-5, 1e-6, "d losses")
    (l_fake + l_real + 0.5 * l_info).backward()
    cmp_grads({k: p.grad for k, p in D.named_parameters()}, g, "dgrad0.", 1e-4, 1e-5)


def test_module_level_api_matches_reference_ops():
    """SocialFeatures / EmbedSocialFeatures / AttentionPooling / EncoderLstm / DecoderFC called one by
    one (the dense, small-batch module API of train.py) against the social_ops golden case."""
    import socialways_amd as sw
    g = golden("social_ops")
    dev = torch.device("cuda:0")
    st = state_from(g, "w0.")
    fe = sw.EmbedSocialFeatures(3, 64, device=dev)
    att = sw.AttentionPooling(64, 64, device=dev)
    fe.load_state_dict(st["feature_embedder"])
    att.load_state_dict(st["attention"])
    obsv, h, sb = torch.from_numpy(g["obsv"]).to(dev), torch.from_numpy(g["h"]).to(dev), g["batches"]
    x4 = sw.get_traj_4d(obsv, [])
    feats = sw.SocialFeatures(x4, sb)
    assert_close(feats.cpu(), g["features"], 1e-5, 1e-6, "dense features")
    emb = fe(feats, sb)
    for s, (a, b) in enumerate(sb):
        assert_close(emb[a:b, a:b].detach().cpu(), g["emb.%d" % s], 1e-5, 2e-6, "emb block %d" % s)
    assert_close(att(emb, h, sb).detach().cpu(), g["S"], 1e-5, 2e-6, "S (autograd path: one workgroup per agent)")
    with torch.no_grad():
        assert_close(att(emb.detach(), h, sb).cpu(), g["S"], 1e-5, 2e-6, "S (no-grad path: one workgroup per scene)")
    # EncoderLstm stand-alone vs the oracle module, sequence then single step, state carried
    enc = sw.EncoderLstm(64, 1, device=dev)
    enc.load_state_dict(st["encoder"])
    oenc = O.EncoderLstm(64, 1)
    oenc.load_state_dict(st["encoder"])
    B = obsv.shape[0]
    enc.init_lstm(torch.zeros(1, B, 64, device=dev), torch.zeros(1, B, 64, device=dev))
    oenc.init_lstm(torch.zeros(1, B, 64), torch.zeros(1, B, 64))
    with torch.no_grad():
        y, yo = enc(x4), oenc(x4.cpu())
        assert_close(y.cpu(), yo, RT, AT, "encoder y (sequence)")
        y1, yo1 = enc(x4[:, -1]), oenc(x4[:, -1].cpu())
        assert_close(y1.cpu(), yo1, RT, AT, "encoder y (single step)")
        assert_close(enc.lstm_h[1].cpu(), oenc.lstm_h[1], RT, AT, "encoder c")
        dec = sw.DecoderFC(160, device=dev)
        dec.load_state_dict(st["decoder"])
        odec = O.DecoderFC(160)
        odec.load_state_dict(st["decoder"])
        torch.manual_seed(5)
        s_, z_ = torch.randn(B, 64), torch.rand(B, 32)
        assert_close(dec(h, s_.to(dev), z_.to(dev)).cpu(), odec(h.cpu(), s_, z_), RT, AT, "decoder")


@pytest.mark.parametrize("B", [37, 2048, 40000])
def test_loss_and_ade_reductions_any_batch_size(B):
    """sw_gan_loss / sw_ade_fde (train.py:484-494, 546-551) against float64 torch sums: the one-workgroup
    path (small B, or no scratch) and the multi-workgroup two-stage path (large B with scratch) agree
    and the multi-workgroup path is run-to-run deterministic."""
    from socialways_amd import _lib as L
    g = torch.Generator().manual_seed(B)
    Tp = 12
    la, lb = torch.rand(B, generator=g).cuda(), torch.rand(B, generator=g).cuda()
    code, z = torch.rand(B, 2, generator=g).cuda(), torch.rand(B, 32, generator=g).cuda()
    tg = torch.tensor([0.05, 0.93]).cuda()
    p4, gt = torch.randn(B, Tp, 4, generator=g).cuda(), torch.randn(B, Tp, 2, generator=g).cuda()
    scratch = torch.zeros(3 * L.RED_BLOCKS).cuda()
    res = []
    for sc in (None, scratch, scratch):
        out = torch.zeros(2, 3).cuda()
        dla, dca, dlb, dcb = torch.zeros(B).cuda(), torch.zeros(B, 2).cuda(), torch.zeros(B).cuda(), torch.ones(B, 2).cuda()
        L.call("sw_gan_loss", L.ptr(la), L.ptr(tg), 0, L.ptr(code), L.ptr(z), L.ptr(lb), 1, B, 0.5, 0.25, L.ptr(out[0]),
               L.ptr(dla), L.ptr(dca), L.ptr(dlb), L.ptr(dcb), L.ptr(sc), L.stream())
        L.call("sw_ade_fde", L.ptr(p4), L.ptr(gt), B, Tp, 0.5, L.ptr(out[1]), L.ptr(sc), L.stream())
        res.append((out.cpu(), dla.cpu(), dca.cpu(), dlb.cpu(), dcb.cpu()))
    d = lambda t: t.double().cpu()
    err = ((d(p4)[..., :2] - d(gt)) * 0.5).norm(dim=-1)
    want = torch.tensor([[((d(la) - 0.05) ** 2).sum(), ((d(code) - d(z)[:, :2]) ** 2).sum(), ((d(lb) - 0.93) ** 2).sum()],
                         [err.sum() / Tp, err[:, -1].sum(), (err ** 2).sum()]])
    for out, dla, dca, dlb, dcb in res:
        assert_close(out.double(), want, 2e-5, 1e-6, "sums")
        assert_close(dla, (la.cpu() - 0.05), 1e-6, 1e-7, "dlabel_a")
        assert_close(dca, 0.5 * (code.cpu() - z.cpu()[:, :2]), 1e-6, 1e-7, "dcode_a")
        assert_close(dlb, (lb.cpu() - 0.93), 1e-6, 1e-7, "dlabel_b")
        assert float(dcb.abs().max()) == 0.0
    assert torch.equal(res[1][0], res[2][0])


@pytest.mark.parametrize("B", [40, 2048])
def test_disc_observation_lstm_precomputed_by_the_decode_launch(B):
    """sw_dec_rollout_fwd_aux runs the discriminator's observation LSTM in idle workgroups of the decode launch;
    sw_disc_fwd(save_lstm=2) then reads the rows: labels, codes and the whole save buffer must equal the plain
    sw_disc_fwd(save_lstm=1) bit for bit, and the rollout itself is unchanged."""
    _d_obs_case(B)


def _d_obs_case(B):
    import socialways_amd as sw
    from socialways_amd import ops, _lib as L
    dev = torch.device("cuda:0")
    torch.manual_seed(B)
    G = sw.Generator(use_social=True, device=dev)
    G.unify()
    D = sw.Discriminator(12, 64, 2, device=dev)
    A = 8
    sb = np.stack([np.arange(B // A) * A, (np.arange(B // A) + 1) * A], axis=1).astype(np.int64)
    scenes = ops.SceneIndex.get(sb, B, dev)
    obsv = torch.randn(B, 8, 2, device=dev).cumsum(1) * 0.1
    noise = torch.rand(B, 32, device=dev)
    real = torch.randn(B, 12, 4, device=dev) * 0.1
    enc, emb, att, dec = G.encoder, G.feature_embedder, G.attention, G.decoder
    ws_a, ws_b = ops.Workspaces(dev), ops.Workspaces(dev)
    pre = ops.d_obs_buffer(ws_a, B, 8, 12)
    assert pre is not None
    ph_a, _ = ops.gen_forward(enc._flat, emb._flat, att._flat, dec._flat, obsv, noise, scenes, 12, True, save=True, ws=ws_a,
                              d_obs=(D._flat, pre))
    ph_b, _ = ops.gen_forward(enc._flat, emb._flat, att._flat, dec._flat, obsv, noise, scenes, 12, True, save=True, ws=ws_b)
    assert torch.equal(ph_a, ph_b)
    la, ca, ctx_a = ops.disc_forward(D._flat, obsv, [ph_a, real], save=True, ws=ws_a, save_lstm=2)
    lb, cb, ctx_b = ops.disc_forward(D._flat, obsv, [ph_b, real], save=True, ws=ws_b, save_lstm=1)
    for x, y in zip(la + ca, lb + cb):
        assert torch.equal(x, y)
    n = L.workspace_floats(L.WS_DSAVE, B, 8, 12, 2)
    assert torch.equal(ctx_a.dsave[:n], ctx_b.dsave[:n])


def _grad_close(a, b, what, rel=2e-4):
    """Gradient tensors: within `rel` of the reference tensor's largest entry (like the predict() gradient tests)."""
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = rel * max(float(np.abs(b).max()), 1e-6)
    err = float(np.abs(a - b).max())
    assert err <= tol, "%s: max|err| %.3e > %.3e (max|ref| %.3e)" % (what, err, tol, float(np.abs(b).max()))


def _pair(hip_mod, oracle_mod):
    oracle_mod.load_state_dict({k: v.detach().cpu().clone() for k, v in hip_mod.state_dict().items()})
    return oracle_mod


def _check_param_grads(hip_mod, oracle_mod, what):
    for (name, p), (_, q) in zip(hip_mod.named_parameters(), oracle_mod.named_parameters()):
        assert p.grad is not None, "%s.%s has no gradient" % (what, name)
        _grad_close(p.grad, q.grad, "%s d/d%s" % (what, name))


def test_standalone_social_modules_are_differentiable_any_scene_size():
    """EmbedSocialFeatures and AttentionPooling called one by one on DENSE tensors, as a user who re-composes the
    reference's modules would (train.py:153-189), with autograd: outputs, input gradients (dense f, h, features) and
    every parameter gradient against the oracle modules under torch's CPU autograd.  Scenes of 5, 1, 70 (above the 64
    agents one workgroup stages in LDS) and 24 agents."""
    import socialways_amd as sw
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    sb = np.array([[0, 5], [5, 6], [6, 76], [76, 100]])
    B = 100
    fe, att = sw.EmbedSocialFeatures(3, 64, device=dev), sw.AttentionPooling(64, 64, device=dev)
    ofe, oatt = _pair(fe, O.EmbedSocialFeatures(3, 64)), _pair(att, O.AttentionPooling(64, 64))
    feats, h = torch.rand(B, B, 3) * 2 - 0.5, torch.randn(B, 64) * 0.5
    wS, wE = torch.randn(B, 64), torch.randn(B, B, 64) * 0.01
    outs = []
    for fe_, att_, d in ((fe, att, dev), (ofe, oatt, torch.device("cpu"))):
        x, hh = feats.to(d).requires_grad_(), h.to(d).requires_grad_()
        emb = fe_(x, sb)
        S = att_(emb, hh, sb)
        ((S * wS.to(d)).sum() + (emb * wE.to(d)).sum()).backward()
        outs.append((emb, S, x.grad, hh.grad))
    (emb, S, dx, dh), (oemb, oS, odx, odh) = outs
    assert_close(emb.detach().cpu(), oemb.detach(), 1e-5, 2e-6, "embedding")
    assert_close(S.detach().cpu(), oS.detach(), 2e-5, 2e-6, "pooled S (scene of 70 agents included)")
    _grad_close(dx, odx, "d/d features")
    _grad_close(dh, odh, "d/d h")
    _check_param_grads(fe, ofe, "feature_embedder")
    _check_param_grads(att, oatt, "attention")
    # the single-agent scene: S = 0 and nothing flows through it (train.py:165)
    assert float(S[5].detach().abs().max()) == 0.0
    # without autograd the same scenes go through the no-grad paths and give the same numbers
    with torch.no_grad():
        S2 = att(fe(feats.to(dev), sb), h.to(dev), sb)
    assert_close(S2.cpu(), S.detach().cpu(), 1e-5, 1e-6, "no-grad path")


def test_standalone_encoder_and_decoder_are_differentiable():
    """EncoderLstm (a sequence, then one more step from the carried state - how predict() uses it, train.py:405,430) and
    DecoderFC with autograd through the stand-alone modules: outputs, the gradients w.r.t. inputs and initial state and
    every parameter gradient against the oracle modules on the CPU."""
    import socialways_amd as sw
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    B, T = 37, 6
    enc, dec = sw.EncoderLstm(64, 1, device=dev), sw.DecoderFC(160, device=dev)
    oenc, odec = _pair(enc, O.EncoderLstm(64, 1)), _pair(dec, O.DecoderFC(160))
    x, h0, c0 = torch.randn(B, T, 4) * 0.5, torch.randn(1, B, 64) * 0.3, torch.randn(1, B, 64) * 0.3
    x1 = torch.randn(B, 4) * 0.5
    w1, w2, w3 = torch.randn(B, T, 64), torch.randn(B, 1, 64), torch.randn(1, B, 64)
    outs = []
    for e_, d in ((enc, dev), (oenc, torch.device("cpu"))):
        xs, x1s = x.to(d).requires_grad_(), x1.to(d).requires_grad_()
        hs, cs = h0.to(d).requires_grad_(), c0.to(d).requires_grad_()
        e_.init_lstm(hs, cs)
        y = e_(xs)
        y1 = e_(x1s)
        ((y * w1.to(d)).sum() + (y1 * w2.to(d)).sum() + (e_.lstm_h[1] * w3.to(d)).sum()).backward()
        outs.append((y, y1, xs.grad, x1s.grad, hs.grad, cs.grad))
    for name, a, b in zip(("y", "y (one more step)"), outs[0][:2], outs[1][:2]):
        assert_close(a.detach().cpu(), b.detach(), RT, AT, "encoder " + name)
    for name, a, b in zip(("x", "x (one more step)", "h0", "c0"), outs[0][2:], outs[1][2:]):
        _grad_close(a, b, "encoder d/d" + name)
    _check_param_grads(enc, oenc, "encoder")
    # DecoderFC
    hh, s, z, wv = torch.randn(B, 64) * 0.5, torch.randn(B, 64) * 0.5, torch.rand(B, 32), torch.randn(B, 2)
    outs = []
    for d_, d in ((dec, dev), (odec, torch.device("cpu"))):
        a, b, c = (t.to(d).requires_grad_() for t in (hh, s, z))
        v = d_(a, b, c)
        (v * wv.to(d)).sum().backward()
        outs.append((v, a.grad, b.grad, c.grad))
    assert_close(outs[0][0].detach().cpu(), outs[1][0].detach(), RT, AT, "decoder v")
    for name, a, b in zip(("h", "s", "z"), outs[0][1:], outs[1][1:]):
        _grad_close(a, b, "decoder d/d" + name)
    _check_param_grads(dec, odec, "decoder")


def test_standalone_modules_at_a_smaller_hidden_size():
    """The module API at `--hidden-size 32` (zero-padded onto the 64-unit kernels): true-shaped inputs / outputs / state
    dicts, forward and autograd of the four sub-modules against the oracle modules built with that size."""
    import socialways_amd as sw
    dev = torch.device("cuda:0")
    H, B, T = 32, 21, 5
    torch.manual_seed(2)
    sb = np.array([[0, 6], [6, 7], [7, 21]])
    fe, att = sw.EmbedSocialFeatures(3, H, device=dev), sw.AttentionPooling(H, H, device=dev)
    enc, dec = sw.EncoderLstm(H, 1, device=dev), sw.DecoderFC(H + H + H // 2, device=dev)
    ofe, oatt = _pair(fe, O.EmbedSocialFeatures(3, H)), _pair(att, O.AttentionPooling(H, H))
    oenc, odec = _pair(enc, O.EncoderLstm(H, 1)), _pair(dec, O.DecoderFC(H + H + H // 2))
    feats, h, x = torch.rand(B, B, 3), torch.randn(B, H) * 0.5, torch.randn(B, T, 4) * 0.5
    z, wS, wy, wv = torch.rand(B, H // 2), torch.randn(B, H), torch.randn(B, T, H), torch.randn(B, 2)
    res = []
    for fe_, att_, enc_, dec_, d in ((fe, att, enc, dec, dev), (ofe, oatt, oenc, odec, torch.device("cpu"))):
        xs, hs = x.to(d).requires_grad_(), h.to(d).requires_grad_()
        enc_.init_lstm(torch.zeros(1, B, H, device=d), torch.zeros(1, B, H, device=d))
        y = enc_(xs)
        assert tuple(y.shape) == (B, T, H) and tuple(enc_.lstm_h[0].shape) == (1, B, H)
        S = att_(fe_(feats.to(d), sb), hs, sb)
        v = dec_(enc_.lstm_h[0].view(B, H), S, z.to(d))
        ((y * wy.to(d)).sum() + (S * wS.to(d)).sum() + (v * wv.to(d)).sum()).backward()
        res.append((y, S, v, xs.grad, hs.grad))
    for name, a, b in zip(("y", "S", "v"), res[0][:3], res[1][:3]):
        assert_close(a.detach().cpu(), b.detach(), 2e-5, 2e-6, name)
    _grad_close(res[0][3], res[1][3], "d/dx")
    _grad_close(res[0][4], res[1][4], "d/dh")
    for m, o, nm in ((fe, ofe, "feature_embedder"), (att, oatt, "attention"), (enc, oenc, "encoder"), (dec, odec, "decoder")):
        for i, ((k, p), (_, q)) in enumerate(zip(m.named_parameters(), o.named_parameters())):
            _grad_close(m.true_view(i, p.grad), q.grad, "%s d/d%s" % (nm, k))


def test_disc_weight_images_are_bit_identical_and_follow_adam():
    """Discriminator passes with registered weight images (sw_disc_images: operand-layout W_hh / W_hh^T, transposed head
    matrices as one LDS-layout block) produce bit-identical outputs, gradients and d/dpred; the Adam update fused into
    the gradient reduction keeps the images equal to a fresh scatter of the updated weights."""
    from socialways_amd import _lib as L
    from socialways_amd import ops
    g = golden("syn_ragged_on")
    G, D, dev = _models(g, 12, True)
    data, obsv, pred, sb, noise = _step_inputs(g)
    obsv, z = obsv.to(dev), noise.to(dev)
    import socialways_amd as sw
    o4, p4 = sw.get_traj_4d(obsv, pred.to(dev))
    fake = torch.from_numpy(g["pred_hat_4d"]).to(dev)
    B = obsv.shape[0]
    lib = L.load()
    n = D._flat.numel()
    tab_h = np.empty((n, 2), dtype=np.int32)
    assert lib.sw_disc_image_table(12, tab_h.ctypes.data) == 0
    tab = torch.from_numpy(tab_h).to(dev)
    img = torch.zeros(lib.sw_disc_image_floats(12), device=dev)
    targets = torch.tensor([0.03, 0.97], device=dev)
    w0 = D._flat.clone()

    def run(with_images):
        D._flat.copy_(w0)
        ws = ops.Workspaces(dev)
        if with_images:
            L.call("sw_disc_images", L.ptr(D._flat), L.ptr(img), L.ptr(tab), 12, L.stream())
        try:
            labels, codes, ctx = ops.disc_forward(D._flat, obsv, [fake, p4], save=True, ws=ws)
            dflat = torch.zeros_like(D._flat)
            m, v = torch.zeros_like(D._flat), torch.zeros_like(D._flat)
            step = torch.ones((), device=dev)
            part = torch.zeros((B + 15) // 16, 3, device=dev)
            ops.disc_backward_gan(D._flat, ctx, labels, codes, targets, (0, 1), z, 1.0 / B, 0.25 / B, dflat, (), ws=ws,
                                  loss_part=part, adam=(m, v, step, 1e-3, 0.9, 0.999, 1e-8))
            dpred = ops.disc_dpred(D._flat, obsv, fake, targets, 1, z, 1.0 / B, 0.25 / B)     # with the UPDATED weights
            torch.cuda.synchronize()
            return [t.clone() for t in labels + codes] + [dflat, D._flat.clone(), dpred, part]
        finally:
            L.call("sw_disc_images", None, None, None, 0, None)

    plain = run(False)
    imaged = run(True)
    for a, b in zip(plain, imaged):
        assert torch.equal(a, b)
    after = img.clone()                       # images as the fused Adam left them
    fresh = torch.zeros_like(img)
    L.call("sw_disc_images", L.ptr(D._flat), L.ptr(fresh), L.ptr(tab), 12, L.stream())
    L.call("sw_disc_images", None, None, None, 0, None)
    torch.cuda.synchronize()
    assert torch.equal(after, fresh)
    assert not torch.equal(w0, D._flat)


def test_adam_packed_kernel_equals_torch_fused_adam():
    """sw_adam_packed (the optimizer step of data-parallel ranks; train.py:379-385) against torch's fused Adam - the op
    behind torch.optim.Adam(fused=True) - on random buffers over several updates: same values up to the placement of
    fused multiply-adds (last bits)."""
    from socialways_amd import _lib as L
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(3)
    n = 27939
    w0 = torch.randn(n, generator=gen).to(dev)
    ws, ms, vs = [w0.clone(), w0.clone()], [torch.zeros(n, device=dev) for _ in range(2)], [torch.zeros(n, device=dev) for _ in range(2)]
    gmax = 0.0
    for t in range(1, 6):
        g = (torch.randn(n, generator=gen) * 10.0 ** float(torch.randint(-6, 1, (1,), generator=gen))).to(dev)
        gmax = max(gmax, float(g.abs().max()))
        step = torch.full((), float(t), device=dev)
        L.call("sw_adam_packed", L.ptr(ws[0]), L.ptr(g), L.ptr(ms[0]), L.ptr(vs[0]), n, L.ptr(step), 1e-3, 0.9, 0.999, 1e-8, 0,
               L.stream())
        torch._fused_adam_([ws[1]], [g], [ms[1]], [vs[1]], [], [step], amsgrad=False, lr=1e-3, beta1=0.9, beta2=0.999,
                           weight_decay=0, eps=1e-8, maximize=False, grad_scale=None, found_inf=None)
    torch.cuda.synchronize()
    # one ulp of the largest term that entered a moment (its entries are sums of terms of both signs), 1e-6 of lr on a weight
    for name, a, b, floor in (("weights", ws[0], ws[1], 1e-9), ("exp_avg", ms[0], ms[1], 2e-7 * gmax),
                              ("exp_avg_sq", vs[0], vs[1], 2e-7 * gmax * gmax)):
        err = (a - b).abs()
        tol = floor + 4e-6 * b.abs()
        assert bool((err <= tol).all()), "%s: max err %.3e" % (name, err.max().item())
    assert float((ws[0] == ws[1]).float().mean()) > 0.95, "bit-identical on almost every element"


@pytest.mark.parametrize("B,obs_pre,To", [(136, False, 8), (2048, False, 8), (40, True, 8), (100, False, 5), (33, True, 3)])
def test_disc_update_in_one_launch_equals_forward_plus_backward(B, obs_pre, To):
    """sw_disc_update (forward + loss gradients + backward of a discriminator update pass per 16-agent tile, the two
    branches' heads side by side on the two wave pairs) against sw_disc_fwd + sw_disc_bwd_gan_adam: labels, codes, the
    reported loss sums, every gradient, the weights and moments after the fused Adam update and the deepcopy snapshot -
    bit for bit (same products in the same order).  To = 8 keeps the LSTM's gates in registers
    between forward and BPTT, other horizons and precomputed observation rows go through the saved rows."""
    import socialways_amd as sw
    from socialways_amd import _lib as L
    from socialways_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(B)
    D = sw.Discriminator(12, 64, 2, device=dev)
    obsv = torch.randn(B, To, 2, device=dev).cumsum(1) * 0.1
    fake, real = torch.randn(B, 12, 4, device=dev) * 0.1, torch.randn(B, 12, 4, device=dev) * 0.1
    z = torch.rand(B, 32, device=dev)
    targets = torch.tensor([0.05, 0.95], device=dev)
    lib = L.load()
    n = D._flat.numel()
    tab_h = np.empty((n, 2), dtype=np.int32)
    assert lib.sw_disc_image_table(12, tab_h.ctypes.data) == 0
    tab = torch.from_numpy(tab_h).to(dev)
    img = torch.zeros(lib.sw_disc_image_floats(12), device=dev)
    w0 = D._flat.clone()
    res = []
    for one_launch in (False, True):
        D._flat.copy_(w0)
        ws = ops.Workspaces(dev)
        L.call("sw_disc_images", L.ptr(D._flat), L.ptr(img), L.ptr(tab), 12, L.stream())
        try:
            assert ops.disc_update_supported(D._flat, B, To, 12)
            if obs_pre:      # the observation LSTM rows as the decode launch leaves them (here: by a plain forward pass)
                ops.disc_forward(D._flat, obsv, [fake, real], save=True, ws=ws, save_lstm=1)
            g = torch.zeros_like(D._flat)
            m, v = torch.zeros_like(D._flat), torch.zeros_like(D._flat)
            step = torch.ones((), device=dev)
            part = torch.zeros((B + 15) // 16, 3, device=dev)
            snap = torch.zeros_like(D._flat)
            adam = (m, v, step, 1e-3, 0.9, 0.999, 1e-8)
            if one_launch:
                labels, codes = ops.disc_update(D._flat, obsv, [fake, real], targets, (0, 1), z, 1.0 / B, 0.25 / B, g, ws,
                                                obs_pre=obs_pre, w_snapshot=snap, loss_part=part, adam=adam)
            else:
                labels, codes, ctx = ops.disc_forward(D._flat, obsv, [fake, real], save=True, ws=ws, save_lstm=2 if obs_pre else 1,
                                                      w_snapshot=snap)
                ops.disc_backward_gan(D._flat, ctx, labels, codes, targets, (0, 1), z, 1.0 / B, 0.25 / B, g, (), ws=ws,
                                      loss_part=part, adam=adam)
            torch.cuda.synchronize()
            res.append([t.clone() for t in labels + codes] + [part, g, D._flat.clone(), m, v, snap, img.clone()])
        finally:
            L.call("sw_disc_images", None, None, None, 0, None)
    names = ["label_fake", "label_real", "code_fake", "code_real", "loss sums", "gradients", "weights", "exp_avg", "exp_avg_sq",
             "snapshot", "images"]
    for name, a, b in zip(names, res[0], res[1]):
        assert torch.equal(a, b), "%s: max |diff| %.3e" % (name, float((a - b).abs().max()))
    assert float(res[1][5].abs().max()) > 0 and not torch.equal(res[1][6], w0)


@pytest.mark.parametrize("R,K,N", [(37, 5, 3), (2048, 48, 32), (130, 160, 80), (16, 64, 256), (1000, 4, 130), (1, 1, 1)])
def test_rows_gemm_on_the_matrix_cores_any_shape_and_stride(R, K, N):
    """sw_rows_gemm (y = x W^T + b with W (N, K), and y (+)= dy W through the transposed strides): the generic-width path's
    matrix product, one wave per 16 rows x 64 columns; shapes that are multiples of nothing, padded row strides, accumulation."""
    from socialways_amd import _lib as L
    g = torch.Generator().manual_seed(R * 131 + K * 7 + N)
    ldx, ldy = K + 3, N + 5
    x = torch.randn(R, ldx, generator=g)
    W = torch.randn(N, K, generator=g) * 0.3
    b = torch.randn(N, generator=g)
    y0 = torch.randn(R, ldy, generator=g)
    xd, Wd, bd = x.cuda(), W.cuda().contiguous(), b.cuda()
    # forward form: w element (k, n) at w[k + n K]
    y = y0.cuda().clone()
    L.call("sw_rows_gemm", L.ptr(xd), ldx, L.ptr(Wd), 1, K, L.ptr(bd), R, K, N, L.ptr(y), ldy, 0, L.stream())
    want = x[:, :K].double() @ W.double().t() + b.double()
    assert_close(y[:, :N].cpu(), want.float(), 2e-5, 2e-6 * max(float(want.abs().max()), 1.0), "y = x W^T + b")
    assert torch.equal(y[:, N:].cpu(), y0[:, N:]), "columns beyond N untouched"
    # backward form, accumulating: dx (+)= dy W, w element (k = n of W, n = k of W) at w[k K + n]
    dy = torch.randn(R, ldy, generator=g)
    dx0 = torch.randn(R, ldx, generator=g)
    dx = dx0.cuda().clone()
    L.call("sw_rows_gemm", L.ptr(dy.cuda()), ldy, L.ptr(Wd), K, 1, None, R, N, K, L.ptr(dx), ldx, 1, L.stream())
    want = dx0[:, :K].double() + dy[:, :N].double() @ W.double()
    assert_close(dx[:, :K].cpu(), want.float(), 2e-5, 2e-6 * max(float(want.abs().max()), 1.0), "dx += dy W")
    assert torch.equal(dx[:, K:].cpu(), dx0[:, K:])


_ENC8_SNIPPET = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
import socialways_amd as sw
from socialways_amd import _lib as L
torch.manual_seed(0)
G = sw.Generator(use_social=True, device="cuda:0")
G.unify()
enc, dec, emb, att = G.encoder._flat, G.decoder._flat, G.feature_embedder._flat, G.attention._flat
lib = L.load()
img = torch.empty(lib.sw_gen_image_floats(), device="cuda")
L.call("sw_gen_images", L.ptr(enc), L.ptr(dec), L.ptr(emb), L.ptr(att), L.ptr(img), L.stream())
out = {}
for B, T in ((2048, 8), (37, 5), (16, 2)):          # the metric shape, a ragged last tile, the shortest sequence
    x = (torch.rand(B, T, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(B)).cumsum(1) * 0.1).contiguous()
    h0, c0 = torch.randn(B, 64, device="cuda") * 0.1, torch.randn(B, 64, device="cuda") * 0.1
    hT, cT = torch.empty(B, 64, device="cuda"), torch.empty(B, 64, device="cuda")
    act, x4s = torch.zeros(T * B * 384, device="cuda"), torch.zeros(T * B * 4, device="cuda")
    L.call("sw_enc_lstm_fwd", L.ptr(x), 0, L.ptr(enc), L.ptr(h0), L.ptr(c0), B, T, L.ptr(hT), L.ptr(cT), None, L.ptr(act), L.ptr(x4s), 0, L.stream())
    torch.cuda.synchronize()
    out[(B, T)] = [t.cpu() for t in (hT, cT, act, x4s)]
torch.save(out, sys.argv[2])
'''


@pytest.mark.gpu
def test_encoder_forward_on_eight_waves_equals_the_four_wave_kernel(tmp_path):
    """enc_lstm_fwd8_kernel (two waves per SIMD, W_hh split 8 ways, rows permuted so that the cell update stays lane-local;
    used up to one tile per CU) against enc_lstm_fwd_kernel: final state, saved rows and saved inputs bit for bit - with an
    initial state, a ragged last tile and a two-step sequence.  The switch is read once per process: two subprocesses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for v in ("0", "1"):
        f = str(tmp_path / ("enc8_%s.pt" % v))
        p = subprocess.run([sys.executable, "-c", _ENC8_SNIPPET, root, f], env=dict(os.environ, SW_ENC8=v), capture_output=True,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[v] = torch.load(f)
    for key in res["0"]:
        for a, b, what in zip(res["0"][key], res["1"][key], ("hT", "cT", "act", "x4s")):
            assert torch.equal(a, b), "B, T = %s: %s differs (max |diff| %.3g)" % (key, what, float((a - b).abs().max()))
    assert float(res["0"][(2048, 8)][0].abs().max()) > 0.0


_FWD2_SNIPPET = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
import socialways_amd as sw
from socialways_amd import _lib as L
torch.manual_seed(0)
G = sw.Generator(use_social=True, device="cuda:0")
G.unify()
enc, dec, emb, att = G.encoder._flat, G.decoder._flat, G.feature_embedder._flat, G.attention._flat
lib = L.load()
img = torch.empty(lib.sw_gen_image_floats(), device="cuda")
L.call("sw_gen_images", L.ptr(enc), L.ptr(dec), L.ptr(emb), L.ptr(att), L.ptr(img), L.stream())
out = {}
for B, To, Tp, social in ((1024, 8, 12, True), (1061, 8, 12, True), (37, 5, 3, False), (16, 2, 1, True), (48, 8, 12, True)):
    gen = torch.Generator(device="cuda").manual_seed(B)
    obsv = (torch.rand(B, To, 2, device="cuda", generator=gen).cumsum(1) * 0.1).contiguous()
    gt = (torch.rand(B, Tp, 2, device="cuda", generator=gen).cumsum(1) * 0.1).contiguous()
    z = torch.rand(B, 32, device="cuda", generator=gen)
    S = torch.randn(B, 64, device="cuda", generator=gen) * 0.3 if social else None
    hT, cT = torch.randn(B, 64, device="cuda", generator=gen) * 0.1, torch.randn(B, 64, device="cuda", generator=gen) * 0.1
    for save in (True, False):
        pred4 = torch.zeros(B, Tp, 4, device="cuda")
        hE, cE = torch.zeros(B, 64, device="cuda"), torch.zeros(B, 64, device="cuda")
        gsave = torch.zeros(L.workspace_floats(L.WS_GSAVE, B, To, Tp), device="cuda") if save else None
        ade = torch.zeros((B + 15) // 16, 3, device="cuda")
        L.call("sw_dec_rollout_fwd", L.ptr(obsv), To, L.ptr(z), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(enc), L.ptr(dec), B, Tp,
               L.ptr(pred4), None if save else L.ptr(hE), None if save else L.ptr(cE), L.ptr(gsave), L.ptr(gt), 0.7, L.ptr(ade),
               L.stream())
        torch.cuda.synchronize()
        keep = [pred4.cpu(), ade.cpu(), hE.cpu(), cE.cpu()]
        if save:
            keep.append(gsave.cpu())      # (zero-filled: the rows the decode loop does not write compare equal)
        out[(B, To, Tp, social, save)] = keep
torch.save(out, sys.argv[2])
'''


@pytest.mark.gpu
def test_decode_forward_on_two_column_blocks_equals_the_16_agent_kernel(tmp_path):
    """dec_rollout_fwd2_kernel (r5 verdict item 3: two 16-agent column blocks per workgroup against every register-resident
    weight operand, selected above 256 tiles) against dec_rollout_fwd_kernel: the prediction, the ADE/FDE partial sums per
    16-agent tile, the final state and every saved row bit for bit - an even and an odd number of tiles with a ragged last
    tile, with and without the pooled social vector / the save buffer, the shortest horizons.  The switch is read once per
    process: two subprocesses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for v in ("0", "1"):
        f = str(tmp_path / ("fwd2_%s.pt" % v))
        p = subprocess.run([sys.executable, "-c", _FWD2_SNIPPET, root, f], env=dict(os.environ, SW_DEC_FWD2=v), capture_output=True,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[v] = torch.load(f)
    for key in res["0"]:
        for i, (a, b) in enumerate(zip(res["0"][key], res["1"][key])):
            assert torch.equal(a, b), "%s: output %d differs (max |diff| %.3g)" % (key, i, float((a - b).abs().max()))
    assert float(res["0"][(1024, 8, 12, True, True)][0].abs().max()) > 0.0
