"""Host-side logic that needs no GPU: the packed Adam wrapper (torch's fused kernel called directly) against
torch.optim.Adam incl. the reference's optimizer state_dict format, and the scene index (pair offsets of small
scenes, row-block records of scenes above 64 agents)."""
import numpy as np
import pytest
import torch

from socialways_amd.ops import SceneIndex
from socialways_amd.trainer import PackedAdam

SHAPES = [(7, 3), (5,), (64, 32), (1,), (33, 4)]


def _packed(shapes):
    offs, o = [], 0
    for s in shapes:
        k = int(np.prod(s))
        offs.append((o, k, s))
        o = (o + k + 3) // 4 * 4                       # every tensor on a 16-byte boundary
    return offs, o


def test_packed_adam_equals_torch_adam_and_reference_state_dict():
    torch.manual_seed(0)
    offs, n = _packed(SHAPES)
    flat, gflat = torch.randn(n), torch.zeros(n)
    for a, k, _ in offs:                               # padding floats are zero and must stay zero
        nxt = min([b for b, _, _ in offs if b > a] + [n])
        flat[a + k:nxt] = 0
    ref = [torch.nn.Parameter(flat[a:a + k].view(s).clone()) for a, k, s in offs]
    opt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999))          # train.py:379-385
    pa = PackedAdam(flat, gflat, offs, 1e-3)
    ext = torch.zeros(())
    for it in range(6):
        for (a, k, s), p in zip(offs, ref):
            g = torch.randn(s)
            p.grad = g.clone()
            gflat[a:a + k] = g.reshape(-1)
        opt.step()
        if it % 2 == 0:
            pa.step()                                  # self-counted
        else:                                          # counter supplied by the caller (the staging kernel's job)
            ext.fill_(float(it + 1))
            pa.t += 1
            pa.step(ext)
    for (a, k, s), p in zip(offs, ref):
        assert torch.equal(flat[a:a + k].view(s), p.detach())
    used = torch.zeros(n, dtype=torch.bool)
    for a, k, _ in offs:
        used[a:a + k] = True
    assert float(flat[~used].abs().max()) == 0.0
    sd, rd = pa.state_dict(), opt.state_dict()
    assert sd["param_groups"][0]["params"] == rd["param_groups"][0]["params"]
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
        assert sd["param_groups"][0][k] == rd["param_groups"][0][k]
    for i in range(len(offs)):
        assert float(sd["state"][i]["step"]) == float(rd["state"][i]["step"]) == 6.0
        assert torch.equal(sd["state"][i]["exp_avg"], rd["state"][i]["exp_avg"])
        assert torch.equal(sd["state"][i]["exp_avg_sq"], rd["state"][i]["exp_avg_sq"])
    # resume: a fresh wrapper loaded from the REFERENCE optimizer's state continues identically
    flat2, g2 = flat.clone(), torch.zeros(n)
    pb = PackedAdam(flat2, g2, offs, 1e-3)
    pb.load_state_dict(rd)
    for (a, k, s), p in zip(offs, ref):
        g = torch.randn(s)
        p.grad = g.clone()
        g2[a:a + k] = g.reshape(-1)
    opt.step()
    pb.step()
    for (a, k, s), p in zip(offs, ref):
        assert torch.equal(flat2[a:a + k].view(s), p.detach())


def test_packed_adam_loads_reference_state_with_gradientless_parameters():
    """The unmodified reference trains with use_social=False (train.py:83): its generator optimizer never sees a
    gradient for the attention / feature-embedder parameters, so torch Adam stores NO state for them
    (train.py:659: `pred_optimizer` has entries 8..21 only).  Loading such a dict must not raise; the
    missing slices keep zero moments and the shared step counter comes from the entries that exist."""
    torch.manual_seed(1)
    offs, n = _packed(SHAPES)
    flat, gflat = torch.randn(n), torch.zeros(n)
    ref = [torch.nn.Parameter(flat[a:a + k].view(s).clone()) for a, k, s in offs]
    opt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999))
    dead = {0, 3}                                      # parameters without gradients
    for it in range(3):
        for i, ((a, k, s), p) in enumerate(zip(offs, ref)):
            p.grad = None if i in dead else torch.randn(s)
        opt.step()
    rd = opt.state_dict()
    assert sorted(rd["state"]) == [1, 2, 4]
    for a, k, s in offs:
        flat[a:a + k] = ref[offs.index((a, k, s))].detach().reshape(-1)
    pa = PackedAdam(flat, gflat, offs, 1e-3)
    pa.load_state_dict(rd)
    assert pa.t == 3
    for i, (a, k, s) in enumerate(offs):
        if i in dead:
            assert float(pa.m[a:a + k].abs().max()) == 0.0 and float(pa.v[a:a + k].abs().max()) == 0.0
        else:
            assert torch.equal(pa.m[a:a + k].view(s), rd["state"][i]["exp_avg"])
    # string keys (a checkpoint that went through json / older torch) load the same way
    pb = PackedAdam(flat.clone(), gflat.clone(), offs, 1e-3)
    pb.load_state_dict({"state": {str(k): v for k, v in rd["state"].items()}, "param_groups": rd["param_groups"]})
    assert pb.t == 3 and torch.equal(pb.m, pa.m) and torch.equal(pb.v, pa.v)
    # the live parameters continue exactly like the reference optimizer
    for i, ((a, k, s), p) in enumerate(zip(offs, ref)):
        g = torch.randn(s)
        p.grad = None if i in dead else g.clone()
        gflat[a:a + k] = 0 if i in dead else g.reshape(-1)
    opt.step()
    pa.step()
    for i, ((a, k, s), p) in enumerate(zip(offs, ref)):
        assert torch.equal(flat[a:a + k].view(s), p.detach()), i


def test_scene_index_small_and_large_scenes():
    sizes = [3, 1, 70, 64, 130]
    ends = np.cumsum(sizes)
    sb = np.stack([ends - sizes, ends], axis=1)
    sc = SceneIndex(sb, int(ends[-1]), "cpu")
    assert sc.S == 5 and sc.amax == 64 and sc.P == 9 + 64 * 64               # single-agent and >64 scenes own no pair rows
    assert sc.pair_off.tolist() == [0, 9, 9, 9, 9 + 4096, 9 + 4096]
    rec = sc.big_blocks.numpy()
    assert sc.NB == 5 + 9 and rec.shape == (14, 8)
    assert rec[:5, 0].tolist() == [2] * 5 and rec[:5, 1].tolist() == [0, 16, 32, 48, 64]
    assert rec[:5, 2].tolist() == [0, 70, 140, 210, 280] and (rec[:5, 3] == 0).all() and (rec[:5, 4] == 5).all()
    assert rec[5, 0] == 4 and rec[5, 2] == 5 * 70 and rec[5, 3] == 350 and rec[5, 4] == 9
    assert sc.big_rows == 5 * 70 + 9 * 130
    with pytest.raises(ValueError):
        SceneIndex(np.array([[0, 3], [4, 6]]), 6, "cpu")                     # must tile [0, B)
    one = SceneIndex([], 9, "cpu")                                           # predict() default: one scene
    assert one.S == 1 and one.amax == 9 and one.NB == 0


@pytest.mark.parametrize("H", [16, 32, 48])
def test_smaller_hidden_sizes_are_zero_padded_with_the_reference_initialisation(H):
    """`--hidden-size` below 64 (train.py:42-44): modules keep kernel-shaped (padded) parameters; the initial weights are
    the ones the reference draws (same construction order, same RNG stream), state_dict() / load_state_dict() speak the
    reference's shapes, the padding is exactly zero, and sizes the kernels cannot hold are refused."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import sw_oracle as O
    import socialways_amd as sw
    torch.manual_seed(1)
    G, D = sw.Generator(H, 1, use_social=True), sw.Discriminator(12, H, 2)
    after = torch.get_rng_state()
    torch.manual_seed(1)
    ref = [O.EncoderLstm(H, 1), O.EmbedSocialFeatures(3, H), O.AttentionPooling(H, H), O.DecoderFC(H + H + H // 2),
           O.Discriminator(12, H, 2)]
    assert torch.equal(after, torch.get_rng_state()), "the generator is left where the reference's construction leaves it"
    for m, o in zip((G.encoder, G.feature_embedder, G.attention, G.decoder, D), ref):
        sd = m.state_dict()
        for k, v in o.state_dict().items():
            assert tuple(sd[k].shape) == tuple(v.shape) and torch.equal(sd[k], v), (type(m).__name__, k)
        for p in o.parameters():
            p.data.mul_(1.5)
        m.load_state_dict(o.state_dict())
        assert all(torch.equal(m.state_dict()[k], v) for k, v in o.state_dict().items())
        assert float((m._flat * (1 - m.pad_mask())).abs().max()) == 0.0
        assert int(m.pad_mask().sum()) == sum(p.numel() for p in o.parameters())
    # optimizer state in the reference's shapes
    G.unify()
    slices = G.packed_slices()
    opt = PackedAdam(G._flat_all, G._gflat_all, slices, 1e-4)
    G._gflat_all.normal_()
    opt.step()
    sd = opt.state_dict()
    order = [p for m in (ref[2], ref[1], ref[0], ref[3]) for p in m.parameters()]      # attention, embedder, encoder, decoder
    assert [tuple(sd["state"][i]["exp_avg"].shape) for i in range(len(order))] == [tuple(p.shape) for p in order]
    opt2 = PackedAdam(torch.zeros_like(G._flat_all), torch.zeros_like(G._gflat_all), slices, 1e-4)
    opt2.load_state_dict(sd)
    mask = torch.cat([torch.nn.functional.pad(m.pad_mask(), (0, (-m.pad_mask().numel()) % 4))
                      for m in (G.attention, G.feature_embedder, G.encoder, G.decoder)])
    assert torch.equal(opt2.m, opt.m * mask) and torch.equal(opt2.v, opt.v * mask)
    for bad in (20, 0):
        with pytest.raises(sw.SocialWaysHipError):
            sw.EncoderLstm(bad, 1)
    from socialways_amd import generic
    assert isinstance(sw.EncoderLstm(128, 1), generic.EncoderLstm)      # above 64 units: the generic-width module
    with pytest.raises(sw.SocialWaysHipError, match="generic"):
        sw.DecoderFC(128 + 128 + 64)


def test_fused_encoder_survives_deepcopy_and_pickle():
    """copy / pickle rebuild a module through cls.__new__(cls) without arguments: the fused EncoderLstm must come back as
    itself (not as the generic-width module its constructor dispatches to for 2 layers / wide sizes), consuming no RNG."""
    import copy
    import io
    import socialways_amd as sw
    from socialways_amd import generic, model
    torch.manual_seed(3)
    enc = model.EncoderLstm(64, 1)
    st = torch.get_rng_state()
    twin = copy.deepcopy(enc)
    assert type(twin) is model.EncoderLstm and torch.equal(torch.get_rng_state(), st)
    assert all(torch.equal(a, b) for a, b in zip(enc.state_dict().values(), twin.state_dict().values()))
    buf = io.BytesIO()
    torch.save(enc, buf)
    buf.seek(0)
    assert type(torch.load(buf, weights_only=False)) is model.EncoderLstm
    g = copy.deepcopy(model.Generator(64, 1, use_social=True))
    assert type(g.encoder) is model.EncoderLstm
    # explicit constructor calls still dispatch: the class default of 2 layers and wide encoders are generic-width modules
    assert type(model.EncoderLstm(64)) is generic.EncoderLstm and type(model.EncoderLstm(128, 1)) is generic.EncoderLstm
    # n_latent_codes given positionally reaches the generic trainer like the keyword does
    t = sw.SocialWaysTrainer.__new__(sw.SocialWaysTrainer, 12, 64, 1e-4, 1e-3, 1, True, True, 0.5, 3)
    assert isinstance(t, generic.GenericTrainer)        # the wide / generic-width family, not the fused 64-unit trainer
