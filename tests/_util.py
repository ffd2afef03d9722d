"""Shared helpers for the parity tests (golden loading, oracle construction)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G_MODS = ("attention", "feature_embedder", "encoder", "decoder")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def state_from(g, prefix):
    """{'encoder': {key: tensor}, ..., 'D': {...}} from flat 'w0.encoder.embed.weight' keys."""
    out = {}
    for k in g.files:
        if k.startswith(prefix):
            mod, key = k[len(prefix):].split(".", 1)
            out.setdefault(mod, {})[key] = torch.from_numpy(np.array(g[k]))
    return out


def as_checkpoint(st, epoch=0):
    return {'epoch': epoch, 'attentioner_dict': st['attention'], 'feature_embedder_dict': st['feature_embedder'],
            'encoder_dict': st['encoder'], 'decoder_dict': st['decoder'], 'D_dict': st['D']}


def dataset_from(g, prefix="ds."):
    return dict(obsvs=g[prefix + "obsvs"], preds=g[prefix + "preds"], batches=g[prefix + "batches"])


def assert_close(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: max|err|=%.3e at %s (got %.8g want %.8g), rtol=%g atol=%g"
                             % (what, err.max(), i, a[i], b[i], rtol, atol))


VARIANT_KW = {          # tests/golden/syn_variants.npz: the reference's module-global switches (train.py:61-69)
    "l2": dict(use_l2_loss=True),
    "variety": dict(use_variety_loss=True),
    "unroll0": dict(n_unrolling_steps=0),
    "unroll2": dict(n_unrolling_steps=2),
    "noinfo": dict(use_info_loss=False),
}


def variant_expected_losses(g, name):
    """The standard 3(U+1)+3 MSE terms of a variant (the variety case recorded 20 extra values - the
    per-k L2 terms of train.py:531 - after them); returns (standard, extras)."""
    v = np.asarray(g[name + ".losses"])
    n = 3 * int(g[name + ".n_d_updates"]) + 3
    return v[:n], v[n:]


def reference_checkpoint(g):
    """The checkpoint dict the unmodified reference wrote (tests/golden/ref_checkpoint.npz, train.py:651-663) rebuilt
    from its arrays: 5 state_dicts + both torch.optim.Adam state dicts (entries only where the reference had them)."""
    ck = {"epoch": int(g["epoch"])}
    for name in ("attentioner_dict", "feature_embedder_dict", "encoder_dict", "decoder_dict", "D_dict"):
        pre = "ck.%s." % name
        ck[name] = {k[len(pre):]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith(pre)}
    for name in ("pred_optimizer", "D_optimizer"):
        lr, b1, b2, eps, wd = [float(x) for x in g["ck.%s.hyper" % name]]
        state = {}
        for i in g["ck.%s.present" % name].tolist():
            state[i] = {"step": torch.tensor(float(g["ck.%s.%d.step" % (name, i)])),
                        "exp_avg": torch.from_numpy(np.array(g["ck.%s.%d.exp_avg" % (name, i)])),
                        "exp_avg_sq": torch.from_numpy(np.array(g["ck.%s.%d.exp_avg_sq" % (name, i)]))}
        ck[name] = {"state": state, "param_groups": [dict(lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, amsgrad=False,
                                                          maximize=False, foreach=None, capturable=False,
                                                          differentiable=False, fused=None,
                                                          params=g["ck.%s.params" % name].tolist())]}
    return ck
