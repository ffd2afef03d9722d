"""Data parallelism of the WIDE trainer (hidden sizes above 64): 2 processes (gloo, sharing the test box's one GPU) train on
scene-aligned shards of each packed batch with three gradient all-reduces per step - eager steps first, then the step as four
captured graph segments around the collectives - and must follow the single-process trajectory (fp32 summation order apart)
with bit-identical replicas."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 128


def _data():
    import socialways_amd as sw
    t = sw.synth_tracks(12, [5, 1, 9, 16, 3, 2, 2, 2, 7, 8, 4, 6], 8, 12, seed=5)
    return sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")


def _epochs(tr, data, n):
    gen = torch.Generator().manual_seed(3)
    out = []
    for e in range(n):
        draws = iter([(0.01 * (e + 1), 0.95, torch.rand(bs, H // 2, generator=gen)) for bs in (65,)])
        ade, fde, losses, sizes = tr.train_epoch(data, 65, draw=lambda bs: next(draws))
        out.append((ade, fde, losses[0].tolist()))
    return out


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                 # different seeds on purpose: construction broadcasts rank 0's replica
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0", process_group=dist.group.WORLD)
    tr.load_checkpoint(torch.load(os.environ["SW_TEST_CK"], map_location="cuda:0"))
    res = _epochs(tr, _data(), 5)
    ret[rank] = (res, tr.gp.flat.double().sum().item(), tr.dp.flat.double().sum().item(), type(tr).__name__,
                 max(len(g["segments"]) for g in tr._graphs.values()))
    tr.release_graphs()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_wide_trainer_two_ranks_follow_the_single_process(tmp_path):
    import torch.multiprocessing as mp
    import socialways_amd as sw
    torch.manual_seed(7)
    one = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0")
    ck = tmp_path / "w0.pt"
    one.save(str(ck), epoch=0)
    os.environ["SW_TEST_CK"] = str(ck)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    ref = _epochs(one, _data(), 5)
    assert ret[0][3] == ret[1][3] == "WideTrainer" and ret[0][4] == 4        # four graph segments around three all-reduces
    assert ret[0][1] == ret[1][1] and ret[0][2] == ret[1][2], "replicas diverged"
    for e in range(5):
        for r in (0, 1):
            ade, fde, losses = ret[r][0][e]
            np.testing.assert_allclose(losses, ref[e][2], rtol=3e-4, atol=1e-6, err_msg="epoch %d rank %d" % (e, r))
            assert abs(ade - ref[e][0]) < 1e-4 * max(1.0, abs(ref[e][0])) and abs(fde - ref[e][1]) < 1e-4 * max(1.0, abs(ref[e][1]))


def test_wide_trainer_l2_and_variety_terms_match_the_oracle():
    """`use_l2_loss` (train.py:525-526) and `use_variety_loss=True` (train.py:527-536 as written: the L2 of agent 19 of the
    packed batch) at 128 units: the 9 MSE terms and every generator gradient of one step against the oracle of that width."""
    import sys
    import socialways_amd as sw
    import sw_oracle as O
    from socialways_amd.wide import WideTrainer
    kw = dict(use_l2_loss=True, use_variety_loss=True, loss_l2_w=0.5)
    torch.manual_seed(7)
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0", **kw)
    assert type(tr) is WideTrainer
    torch.manual_seed(7)
    orc = O.SocialWaysOracle(12, hidden_size=H, use_social=True, **kw)
    data = _data()
    B, sb = 36, data.the_batches[:6]
    noise = torch.rand(B, H // 2, generator=torch.Generator().manual_seed(2))
    rec = {}
    out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    got = tr.losses_from(out, [B], 12, data.ss)[0]
    want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.03, 0.94, noise, data.ss, record=rec)
    # the reported g_l2 slot: the oracle reports the l2 term, the wide trainer's row keeps it at 0 like SocialWaysTrainer.step
    np.testing.assert_allclose(np.asarray(got)[[0, 1, 2, 3, 4, 5, 7, 8]], np.asarray(want)[[0, 1, 2, 3, 4, 5, 7, 8]], rtol=5e-5, atol=3e-6)
    for name in ("attention", "feature_embedder", "encoder", "decoder"):
        for k, p in getattr(tr.G, name).named_parameters():
            w = rec["g_grads"][name + "." + k]
            err = float((p.grad.cpu() - w).abs().max())
            assert err <= 2e-4 * max(float(w.abs().max()), 1e-12) + 1e-9, (name, k, err)


def test_wide_trainer_workspace_eviction_keeps_the_trajectory():
    """Ragged data produces many batch shapes; the wide trainer keeps buffers + captured graphs for a few of them (LRU) and
    re-creates the rest: the trajectory over alternating shapes must not depend on the cap."""
    import socialways_amd as sw
    data = _data()
    sbs = [(data.the_batches[:k], int(data.the_batches[k - 1][1])) for k in (6, 4, 9)]
    res = []
    for cap in (6, 1):
        torch.manual_seed(7)
        tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0")
        tr.MAX_WORKSPACES = cap
        gen = torch.Generator().manual_seed(4)
        outs = []
        for it in range(12):                       # every shape four times: eager, eager, capture, replay (or rebuilt, cap = 1)
            sb, B = sbs[it % 3]
            outs.append(tr.step(data.obsv[:B], data.pred[:B], sb, 0.02, 0.95, torch.rand(B, H // 2, generator=gen), data.ss).cpu())
        res.append((torch.stack(outs), tr.gp.flat.clone(), tr.dp.flat.clone(), len(tr._ws)))
    assert res[0][3] == 3 and res[1][3] == 1
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_wide_trainer_checkpoint_with_another_lr_is_not_replayed_from_old_graphs():
    """A captured wide step bakes the optimizers' host scalars (lr, betas, eps) in.  Loading a checkpoint whose Adam dicts
    carry another lr after a layout was captured must re-capture: the graph-replaying trainer against an eager one
    through the same sequence - capture at lr0, load a checkpoint with 3 x lr, four more steps - bit for bit."""
    import copy
    import socialways_amd as sw
    data = _data()
    sb, B = data.the_batches[:6], int(data.the_batches[5][1])
    res = []
    for use_graph in (True, False):
        torch.manual_seed(7)
        tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0", use_graph=use_graph)
        gen = torch.Generator().manual_seed(4)
        outs = []

        def run(n):
            for it in range(n):
                outs.append(tr.step(data.obsv[:B], data.pred[:B], sb, 0.02, 0.95, torch.rand(B, H // 2, generator=gen), data.ss).cpu())
        run(4)                                             # eager, eager, capture, replay
        ck = copy.deepcopy(tr.checkpoint(1))
        for key in ("pred_optimizer", "D_optimizer"):
            for g in ck[key]["param_groups"]:
                g["lr"] = 3.0 * g["lr"]
        tr.load_checkpoint(ck)
        assert tr.D_optimizer.param_groups[0]["lr"] == pytest.approx(3e-3)
        run(4)
        res.append((torch.stack(outs), tr.gp.flat.clone(), tr.dp.flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("use_social,n_next", [(False, 12), (True, 1), (True, 3)])
def test_wide_trainer_edge_configurations_match_the_generic_path(use_social, n_next):
    """No social block (train.py:83's default), a one-step horizon (no re-fed encoder step at all) and a short one: the wide
    engine against the layer-by-layer path under torch's tape on the same modules - sums, rollout, every gradient."""
    import socialways_amd as sw
    from socialways_amd.generic import GenericTrainer
    from socialways_amd.wide import WideTrainer
    torch.manual_seed(5)
    a = WideTrainer(n_next, hidden_size=H, use_social=use_social, device="cuda:0", use_graph=False)
    torch.manual_seed(5)
    b = GenericTrainer(n_next, hidden_size=H, use_social=use_social, device="cuda:0")
    t = sw.synth_tracks(6, [5, 1, 9, 16, 3, 2], 8, n_next, seed=5)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    B, sb = 36, data.the_batches[:6]
    noise = torch.rand(B, H // 2, generator=torch.Generator().manual_seed(2))
    ra = a.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    rb = b.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    np.testing.assert_allclose(ra.cpu().numpy(), rb.cpu().numpy(), rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(a.last_pred_hat.cpu().numpy(), b.last_pred_hat.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for (n, p), (_, q) in zip(a.G.named_parameters(), b.G.named_parameters()):     # (the generic trainer drops D's gradients at the end)
        if q.grad is None:            # torch's tape leaves unused parameters without a gradient; the engine writes zeros
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        err = float((p.grad - q.grad).abs().max())
        assert err <= 1e-4 * max(float(q.grad.abs().max()), 1e-12) + 1e-10, (n, err)
    for (n, p), (_, q) in zip(a.D.named_parameters(), b.D.named_parameters()):     # D after its two updates and the Linear-only restore
        d = (p.detach() - q.detach()).abs()
        assert float(d.max()) <= 4.4e-3 and float((d <= 1e-4).float().mean()) > 0.98, n


def test_wide_trainer_refuses_scenes_above_the_limit_even_when_no_small_scene_has_pairs():
    """Scenes above 64 agents are not supported on the wide path - also when they are the ONLY multi-agent scenes of the batch
    (no in-scene pair of a small scene: the pair count the social block is gated on is then 0; found by the randomised sweep:
    the step ran without social pooling instead of refusing)."""
    import socialways_amd as sw
    t = sw.synth_tracks(4, [70, 1, 1, 2], 8, 12, seed=5)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0")
    B, sb = 72, data.the_batches[:3]
    with pytest.raises(sw.SocialWaysHipError, match="above 64 agents"):
        tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, torch.rand(B, H // 2), data.ss)
    tr2 = sw.SocialWaysTrainer(12, hidden_size=H, use_social=False, device="cuda:0")      # without the social block: fine
    assert torch.isfinite(tr2.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, torch.rand(B, H // 2), data.ss)).all()
