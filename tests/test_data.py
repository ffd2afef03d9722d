"""Host-side data glue against the reference's own outputs: toy generator, normalisation, 4/5 split,
greedy scene packing, scene-aligned sharding."""
import numpy as np
import torch

import sw_oracle as O
from _util import golden


def test_toy_generator_reproduces_create_toy():
    import socialways_amd as sw
    for n_cond, name in ((8, "toy_768_8_3"), (6, "toy_768_6_3")):
        g = golden(name)
        t = sw.toy_tracks(768, n_cond, 3)
        assert np.array_equal(t["batches"], g["batches"])
        assert np.array_equal(t["times"], g["times"])
        assert np.allclose(t["obsvs"], g["obsvs"], rtol=0, atol=1e-7)
        assert np.allclose(t["preds"], g["preds"], rtol=0, atol=1e-7)


def test_dataset_normalisation_split_and_packing():
    import socialways_amd as sw
    g = golden("toy_b64_on")
    toy = golden("toy_768_8_3")
    d = sw.SceneDataset(toy["obsvs"], toy["preds"], toy["batches"], toy["times"], device="cpu")
    o = O.load_and_normalise(toy["obsvs"], toy["preds"], toy["batches"])
    assert d.ss == o["ss"] == float(g["ss"])
    assert d.train_size == o["train_size"] == int(g["train_size"])
    assert d.n_train_samples == int(g["n_train_samples"])
    assert torch.equal(d.obsv, o["obsv"]) and torch.equal(d.pred, o["pred"])
    steps = list(d.packed_steps(64))
    assert [b - a for a, b, _ in steps] == g["step_agents"].tolist()           # train.py:446-456
    for a, b, sb in steps:
        assert sb[0, 0] == 0 and sb[-1, 1] == b - a and (sb[1:, 0] == sb[:-1, 1]).all()
    # denormalise(normalise(x)) == x
    x = toy["obsvs"].astype(np.float32).copy()
    assert np.allclose(d.scale.denormalize(d.obsv.numpy()), x, atol=1e-5)


def test_int16_batches_are_widened():
    import socialways_amd as sw
    t = sw.synth_tracks(6, 4)
    d = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"].astype(np.int16), device="cpu")   # parse_utils.py:490
    assert d.the_batches.dtype == np.int64 and d.n_train_samples == 16


def test_single_scene_dataset_edge_case():
    import socialways_amd as sw
    t = sw.synth_tracks(1, 5)
    d = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cpu")
    assert d.n_test_samples == 1 and len(d.the_batches) == 2                   # train.py:107-109


def test_shard_scenes_properties():
    import socialways_amd as sw
    rng = np.random.default_rng(0)
    for _ in range(50):
        n = rng.integers(1, 65, size=rng.integers(1, 40))
        ends = np.cumsum(n)
        sb = np.stack([ends - n, ends], 1)
        for w in (1, 2, 4, 8):
            sh = sw.shard_scenes(sb, w)
            assert len(sh) == w and sh[0][0] == 0 and sh[-1][1] == len(sb)
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:])) and all(lo <= hi for lo, hi in sh)
    sb = np.stack([np.arange(256) * 8, np.arange(1, 257) * 8], 1)
    assert [hi - lo for lo, hi in sw.shard_scenes(sb, 8)] == [32] * 8           # uniform scenes split evenly


def test_synth_tracks_match_oracle_generator():
    import socialways_amd as sw
    a, b = sw.synth_tracks(5, [3, 1, 8, 2, 6], seed=7), O.synth_dataset(5, [3, 1, 8, 2, 6], seed=7)
    for k in ("obsvs", "preds", "batches", "times"):
        assert np.array_equal(a[k], b[k])


def test_biwi_pipeline_matches_reference_parser(tmp_path):
    """obsmat.txt -> windows -> scenes (create_dataset.py + utils/parse_utils.py:231-320,457-508): the
    golden arrays were produced by the reference's BIWIParser + create_dataset on this synthetic file."""
    import socialways_amd as sw
    from socialways_amd import data as D
    g = golden("biwi_synth")
    path = str(tmp_path / "obsmat.txt")
    D.write_biwi_obsmat(path, g["frames"], g["ids"], g["pos"], g["vel"])
    p_data, t_data, interval = D.parse_biwi(path)
    assert interval == int(g["interval"]) == 6
    obsvs, preds, times, batches = D.biwi_to_npz(path, str(tmp_path / "data-8-12.npz"))
    assert np.array_equal(obsvs, g["obsvs"]) and np.array_equal(preds, g["preds"])
    assert list(times) == g["times"].tolist() and np.array_equal(batches, g["batches"])
    assert obsvs.shape[1:] == (8, 2) and preds.shape[1:] == (12, 2)
    d = sw.SceneDataset.from_npz(str(tmp_path / "data-8-12.npz"), device="cpu")     # train.py:89-124 on that file
    assert d.n_past == 8 and d.n_next == 12 and d.train_size == (len(batches) * 4) // 5
