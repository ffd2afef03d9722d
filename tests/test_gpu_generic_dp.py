"""The GENERIC-WIDTH trainer (widths that are not multiples of 32, e.g. 80 units: layer by layer under torch's tape) with
the options the other engines have: data parallelism - 2 processes (gloo, sharing the test box's one GPU) on scene-aligned
shards must follow the single process with identical replicas - and the reference's L2 / variety terms against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 80


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


def _flat(tr):
    return torch.cat([p.detach().double().reshape(-1) for p in list(tr.G.parameters()) + list(tr.D.parameters())])


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                 # different seeds on purpose: construction broadcasts rank 0's replica
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0", process_group=dist.group.WORLD,
                              use_l2_loss=True, use_variety_loss=True)
    tr.load_checkpoint(torch.load(os.environ["SW_TEST_CK"], map_location="cuda:0"))
    res = _epochs(tr, _data(), 3)
    ret[rank] = (res, _flat(tr).cpu(), type(tr).__name__)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_generic_trainer_two_ranks_follow_the_single_process(tmp_path):
    import torch.multiprocessing as mp
    import socialways_amd as sw
    torch.manual_seed(7)
    one = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0", use_l2_loss=True, use_variety_loss=True)
    ck = tmp_path / "g0.pt"
    one.save(str(ck), epoch=0)
    os.environ["SW_TEST_CK"] = str(ck)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    ref = _epochs(one, _data(), 3)
    assert ret[0][2] == ret[1][2] == "GenericTrainer"
    assert torch.equal(ret[0][1], ret[1][1]), "replicas diverged"
    w1 = _flat(one).cpu()
    assert float((ret[0][1] - w1).abs().max()) <= 2e-5, "two ranks left the single-process trajectory"
    for e in range(3):
        for r in (0, 1):
            ade, fde, losses = ret[r][0][e]
            np.testing.assert_allclose(losses, ref[e][2], rtol=3e-4, atol=1e-6, err_msg="epoch %d rank %d" % (e, r))
            assert abs(ade - ref[e][0]) < 1e-4 * max(1.0, abs(ref[e][0])) and abs(fde - ref[e][1]) < 1e-4 * max(1.0, abs(ref[e][1]))


def test_generic_trainer_l2_and_variety_terms_match_the_oracle():
    """`use_l2_loss` (train.py:525-526) and `use_variety_loss=True` (train.py:527-536 as written: the L2 of agent 19 of the
    packed batch) at 80 units: the MSE terms and every generator gradient of one step against the oracle of that width."""
    import socialways_amd as sw
    import sw_oracle as O
    from socialways_amd.generic import GenericTrainer
    kw = dict(use_l2_loss=True, use_variety_loss=True, loss_l2_w=0.5)
    torch.manual_seed(7)
    tr = sw.SocialWaysTrainer(12, hidden_size=H, use_social=True, device="cuda:0", **kw)
    assert type(tr) is GenericTrainer
    torch.manual_seed(7)
    orc = O.SocialWaysOracle(12, hidden_size=H, use_social=True, **kw)
    data = _data()
    B, sb = 36, data.the_batches[:6]
    noise = torch.rand(B, H // 2, generator=torch.Generator().manual_seed(2))
    rec = {}
    out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    got = tr.losses_from(out, [B], 12, data.ss)[0]
    want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.03, 0.94, noise, data.ss, record=rec)
    np.testing.assert_allclose(np.asarray(got)[[0, 1, 2, 3, 4, 5, 7, 8]], np.asarray(want)[[0, 1, 2, 3, 4, 5, 7, 8]], rtol=5e-5, atol=3e-6)
    for name in ("attention", "feature_embedder", "encoder", "decoder"):
        for k, p in getattr(tr.G, name).named_parameters():
            w = rec["g_grads"][name + "." + k]
            err = float((p.grad.cpu() - w).abs().max())
            assert err <= 2e-4 * max(float(w.abs().max()), 1e-12) + 1e-9, (name, k, err)


@pytest.mark.parametrize("width", [128, 80])
def test_wider_trainers_evaluate_like_the_oracle(width):
    """test() (train.py:563-616) of the wide (128 units) and the generic-width (80) trainer: K = 4 sampled futures per
    held-out scene from the same generator state as the oracle of that width - min / avg ADE and FDE."""
    import socialways_amd as sw
    import sw_oracle as O
    t = sw.synth_tracks(10, [5, 1, 9, 16, 3, 2, 2, 7, 4, 6], 8, 12, seed=9)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    odata = O.load_and_normalise(t["obsvs"], t["preds"], t["batches"])
    torch.manual_seed(11)
    tr = sw.SocialWaysTrainer(12, hidden_size=width, use_social=True, device="cuda:0")
    torch.manual_seed(11)
    orc = O.SocialWaysOracle(12, hidden_size=width, use_social=True)
    assert type(tr).__name__ == ("WideTrainer" if width == 128 else "GenericTrainer")
    torch.manual_seed(123)
    got = tr.test(data, n_gen_samples=4)
    torch.manual_seed(123)
    want = orc.test(odata, n_gen_samples=4)
    np.testing.assert_allclose(np.asarray(got), np.asarray(want), rtol=1e-4, atol=1e-6)
