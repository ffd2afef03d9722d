"""Toy statistics (calc_statistics.py compute_1nn / compute_wasserstein) on the GPU against
the reference's golden values and, end to end, on prediction files written by SocialWaysTrainer.test()."""
import os

import numpy as np
import pytest
import torch

import sw_oracle as O
from _util import golden, assert_close

pytestmark = pytest.mark.gpu


def test_distance_matrix_matches_numpy():
    import socialways_amd as sw
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(9, 5, 7, 2)).astype(np.float32), rng.normal(size=(4, 5, 7, 2)).astype(np.float32)
    D = sw.stats.traj_dist(a, b, 3).cpu().numpy()
    want = np.sqrt(((a[:, None, :, 3:] - b[None, :, :, 3:]).astype(np.float64) ** 2).sum(-1)).mean(-1).transpose(2, 0, 1)
    assert D.shape == (5, 9, 4)
    assert_close(D, want, 2e-6, 1e-7, "mean displacement matrix")
    Dm = sw.stats.traj_dist(a, a, 3)
    assert torch.equal(Dm, Dm.transpose(1, 2)), "d(i,j) == d(j,i) bitwise (the reference mirrors i<j)"


def test_1nn_and_emd_match_reference():
    import socialways_amd as sw
    g = golden("toy_stats")
    for i in range(len(g["sigmas"])):
        assert_close(sw.stats.compute_1nn(g["real"], g["fake.%d" % i]), g["one_nn.%d" % i], 0, 1e-12, "1nn %d" % i)
        assert abs(sw.stats.compute_wasserstein(g["real"], g["fake.%d" % i]) - float(g["emd.%d" % i])) < 1e-6
    assert_close(sw.stats.compute_1nn(g["g.real"], g["g.fake"], 3), g["g.one_nn"], 0, 1e-12, "1nn generic")
    assert abs(sw.stats.compute_wasserstein(g["g.real"], g["g.fake"], 3) - float(g["g.emd"])) < 1e-6
    with pytest.raises(ValueError):
        sw.stats.compute_wasserstein(g["real"], g["fake.0"][:7])


def test_statistics_of_written_predictions(tmp_path):
    """test(write_to_file=...) on the 6-condition toy set -> calc_and_store_stats, compared with the
    oracle's statistics of the same prediction files (the pipeline calc_statistics.py:70-125 runs)."""
    import socialways_amd as sw
    toy = golden("toy_768_6_3")
    data = sw.SceneDataset(toy["obsvs"], toy["preds"], toy["batches"], toy["times"], device="cuda:0")
    torch.manual_seed(0)
    tr = sw.SocialWaysTrainer(2, use_social=True, device="cuda:0")
    tr.epoch = 5
    d = tmp_path / "model" / "5"
    tr.test(data, n_gen_samples=20, write_to_file=str(d))
    files = sorted(os.listdir(d))
    assert len(files) == len(data.test_batches)
    real = np.concatenate((toy["obsvs"], toy["preds"]), axis=1).reshape((-1, 6, 4, 2))[:20]
    s1, sw_ = sw.stats.calc_and_store_stats(str(tmp_path / "model"), real, 2, 2, stats_file=str(tmp_path / "stats20.npz"))
    a1 = aw = 0.0
    for f in files:
        z = np.load(d / f)
        fake = np.concatenate((np.broadcast_to(z["obsvs"][None], (20, 6, 2, 2)), z["preds_our"][:20]), axis=2).astype(np.float32)
        a1 += O.compute_1nn(real, fake)[0]
        aw += O.compute_wasserstein(real, fake)
    assert list(s1) == [5] and abs(s1[5] - a1 / len(files)) < 1e-12 and abs(sw_[5] - aw / len(files)) < 1e-6
    st = np.load(tmp_path / "stats20.npz")
    assert st["stats_1nn"].shape == (1,) and abs(st["stats_wst"][0] - sw_[5]) < 1e-12


def test_end_to_end_example(tmp_path):
    """examples/train_toy.py: toy data -> epochs of training -> test() + npz -> checkpoint -> statistics;
    the losses stay finite, ADE improves over the first epochs and every artefact is written."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_toy", os.path.join(root, "examples", "train_toy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tr, s1, emd = mod.main(["--epochs", "6", "--test-every", "3", "--n-samples", "384", "--out", str(tmp_path)])
    assert sorted(s1) == [3, 6] and all(np.isfinite(list(emd.values()))) and all(0.0 <= v <= 1.0 for v in s1.values())
    ck = torch.load(tmp_path / "toy.pt", weights_only=False)
    assert sorted(ck) == sorted(['epoch', 'attentioner_dict', 'feature_embedder_dict', 'encoder_dict', 'decoder_dict',
                                 'pred_optimizer', 'D_dict', 'D_optimizer'])
    assert len(os.listdir(tmp_path / "preds" / "6")) > 0 and (tmp_path / "stats20.npz").exists()
