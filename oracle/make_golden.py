"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference, which does not exist on the GPU box).
The reference `train.py` is a script (argparse + data load + model construction + epoch loop at
import time, unconditional .cuda(), time.clock) - SURVEY.md §0.3.  It is imported here with the
CPU shims of SURVEY.md §8c and driven through its own `train()`, `predict()`, `D.forward()`,
`SocialFeatures()`, `test()` ...; nothing of its source is copied.  Outputs are data only
(inputs + expected outputs) - the reference's text never enters this repo.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz  (~1 min)

Cases (all fp32, torch CPU; `threads` recorded in every file)
  toy_768_8_3 / toy_768_6_3
        the toy datasets, produced by running the reference's create_toy.py as __main__ under
        the numpy-2 shim of SURVEY.md §0.7 (np.random.rand(1) -> python float, same RNG stream).
  toy_b64_off / toy_b64_on
        toy (768,8,3), --batch-size 64, seeds torch=0 / numpy=0, use_social off/on, ONE epoch of
        the reference's train(): the 10x9 MSE terms, (agents, scenes) of every step, label-noise
        scalars and z of every step, epoch ADE/FDE, weights before and after the epoch.
  syn_s16a8_off / syn_s16a8_on / syn_ragged_on
        synthetic 8+12 tracks (SURVEY.md §8d generator), one packed train() step with every
        intermediate: hT, S, pred_hat_4d, d(g_loss)/d(pred_hat_4d), D outputs, all D gradients of
        both D updates, all G gradients, weights before/after.
  social_ops
        SocialFeatures -> EmbedSocialFeatures -> AttentionPooling called directly on a ragged batch,
        plus the reference's scalar DCA()/Bearing() per-pair specification.
  test_eval
        test(n_gen_samples=4, write_to_file=...) on 2 held-out scenes: K predictions, the four
        metrics and the prediction-npz arrays (schema of train.py:598-599).
  syn_big_on
        like syn_ragged_on with scenes of 70 and 90 agents (above the 64-agent limit of the one-workgroup-per-scene
        kernels): one packed train() step of the reference with every intermediate and gradient.
  toy_multi
        toy (768,8,3), --batch-size 64, use_social on, FIVE consecutive epochs of train() in one process: per-epoch
        ADE/FDE and MSE terms plus every RNG draw, to check that parity holds beyond the first epoch.
  syn_variants
        the syn_s16a8_on step again with the reference's loss / unrolling switches flipped (module
        globals of train.py:61-69): use_l2_loss, use_variety_loss (as written in train.py:527-536),
        n_unrolling_steps 0 and 2, use_info_loss off.  Per variant: every MSE value train() computed,
        the D gradients of the last D update, the G gradients, D's weights after the step.
  toy_stats
        compute_1nn / compute_wasserstein of calc_statistics.py (the script's module body - dataset load,
        plotting - is not run: only its import statements and these two function definitions are compiled,
        from the file where it lies) on K=20 real toy samples vs perturbed copies, several noise levels.
  biwi_synth
        a synthetic BIWI-format obsmat.txt (no ETH/UCY data exists here) through the
        reference's BIWIParser + create_dataset (utils/parse_utils.py): windows, times, scene batches.
"""
import contextlib
import glob
import importlib.util
import io
import os
import runpy
import shutil
import sys
import tempfile
import time

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
import sw_oracle  # noqa: E402  (synthetic track generator only, so inputs are the tests' inputs)

_counter = [0]


def import_reference(dataset, batch_size, seed=0, use_social=False, epochs=0, checkpoint_from=None):
    """Import /root/reference/train.py on CPU (SURVEY.md §8c recipe).  `epochs` > 0 lets the module-level loop of
    train.py:646-668 run (training, its own torch.save at epoch 50, its own test() every 5 epochs);
    `checkpoint_from` = a .pt the reference wrote earlier, placed where train.py:56,622 looks for it."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    time.clock = time.perf_counter
    root = tempfile.mkdtemp(prefix="swref_")
    work = os.path.join(root, "work")
    os.makedirs(work)
    np.savez(os.path.join(root, "hotel-8-12.npz"), **dataset)
    os.makedirs(os.path.join(root, "trained_models"))
    if checkpoint_from is not None:
        shutil.copy(checkpoint_from, os.path.join(root, "trained_models", "socialWays-hotel.pt"))
    old_cwd, old_argv, old_path = os.getcwd(), sys.argv, list(sys.path)
    os.chdir(work)
    sys.argv = ["train.py", "--epochs", str(epochs), "--batch-size", str(batch_size)]
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    torch.manual_seed(seed)
    np.random.seed(seed)
    _counter[0] += 1
    spec = importlib.util.spec_from_file_location("sw_ref_train_%d" % _counter[0], os.path.join(REF, "train.py"))
    m = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        spec.loader.exec_module(m)
    sys.argv, sys.path = old_argv, old_path
    if epochs == 0:
        m.epoch = 1
    m.use_social = use_social
    m._root = root

    def cleanup():
        os.chdir(old_cwd)
        shutil.rmtree(root, ignore_errors=True)
    m._cleanup = cleanup
    return m


def toy_dataset(n_samples, n_conditions, n_modes):
    """Run the reference's create_toy.py as __main__ (its packing sits under __main__)."""
    root = tempfile.mkdtemp(prefix="swtoy_")
    path = os.path.join(root, "toy.npz")
    real_rand, old_argv = np.random.rand, sys.argv
    np.random.rand = lambda *a: float(real_rand(*a)[0]) if a == (1,) else real_rand(*a)
    sys.argv = ["create_toy.py", "--npz", path, "--n_samples", str(n_samples),
                "--n_conditions", str(n_conditions), "--n_modes", str(n_modes)]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            runpy.run_path(os.path.join(REF, "create_toy.py"), run_name="__main__")
        d = dict(np.load(path))
    finally:
        np.random.rand, sys.argv = real_rand, old_argv
        shutil.rmtree(root, ignore_errors=True)
    return d


def flat_state(m, prefix=""):
    out = {}
    for name, mod in (("attention", m.attention), ("feature_embedder", m.feature_embedder),
                      ("encoder", m.encoder), ("decoder", m.decoder), ("D", m.D)):
        for k, v in mod.state_dict().items():
            out["%s%s.%s" % (prefix, name, k)] = v.detach().clone().numpy()
    return out


class Recorder:
    """Wraps the reference module's globals to expose what train()/test() keep local."""

    def __init__(self, m):
        self.m = m
        self.mse, self.uniform, self.noise = [], [], []
        self.predicts, self.dec_first, self.d_calls = [], [], []
        self.d_grads, self.g_grads, self.dpred = [], [], []
        self._in_predict = False
        real_mse = m.mse_loss
        m.mse_loss = lambda a, b: self._rec(self.mse, real_mse(a, b))
        real_predict = m.predict

        def predict(*a, **k):
            self._first = True
            out = real_predict(*a, **k)
            self.predicts.append(out.detach().clone())
            return out
        m.predict = predict
        real_dec = m.decoder.forward

        def dec_forward(h, s, z):
            if getattr(self, "_first", False):
                self.dec_first.append((h.detach().clone(), s.detach().clone()))
                self._first = False
            return real_dec(h, s, z)
        m.decoder.forward = dec_forward
        real_D = m.D.forward

        def D_forward(obsv, pred):
            if pred.requires_grad:
                pred.register_hook(lambda g: self.dpred.append(g.detach().clone()))
            label, code = real_D(obsv, pred)
            self.d_calls.append((obsv.detach().clone(), pred.detach().clone(),
                                 label.detach().clone(), code.detach().clone()))
            return label, code
        m.D.forward = D_forward
        real_dstep, real_gstep = m.D_optimizer.step, m.predictor_optimizer.step

        def dstep(*a, **k):
            self.d_grads.append({k_: (np.zeros(tuple(p.shape), np.float32) if p.grad is None     # info loss off: the
                                      else p.grad.detach().clone().numpy())                       # code head gets none
                                 for k_, p in m.D.named_parameters()})
            return real_dstep(*a, **k)

        def gstep(*a, **k):
            g = {}
            for name, mod in (("attention", m.attention), ("feature_embedder", m.feature_embedder),
                              ("encoder", m.encoder), ("decoder", m.decoder)):
                for k_, p in mod.named_parameters():
                    g[name + "." + k_] = (np.zeros(tuple(p.shape), np.float32) if p.grad is None
                                          else p.grad.detach().clone().numpy())
            self.g_grads.append(g)
            return real_gstep(*a, **k)
        m.D_optimizer.step, m.predictor_optimizer.step = dstep, gstep
        self._real_uniform, self._real_rand = np.random.uniform, torch.rand

        def uniform(*a, **k):
            v = self._real_uniform(*a, **k)
            self.uniform.append(v)
            return v

        def rand(*a, **k):
            v = self._real_rand(*a, **k)
            self.noise.append(v.clone())
            return v
        np.random.uniform, torch.rand = uniform, rand

    @staticmethod
    def _rec(lst, v):
        lst.append(float(v.item()))
        return v

    def close(self):
        np.random.uniform, torch.rand = self._real_uniform, self._real_rand


def run_epoch(dataset, batch_size, use_social, seed=0, overrides=None):
    m = import_reference(dataset, batch_size, seed, use_social)
    for k, v in (overrides or {}).items():      # loss / unrolling switches are module globals read by train()
        assert hasattr(m, k), k
        setattr(m, k, v)
    w0 = flat_state(m, "w0.")
    rec = Recorder(m)
    buf = io.StringIO()
    np.random.seed(seed)          # import consumed nothing from numpy; keep the stream explicit
    with contextlib.redirect_stdout(buf):
        m.train()
    rec.close()
    w1 = flat_state(m, "w1.")
    n_steps = len(rec.g_grads)
    per = 3 * (m.n_unrolling_steps + 1) + 3 + (20 if m.use_variety_loss else 0)
    losses = np.asarray(rec.mse, np.float64).reshape(n_steps, per)
    # ADE/FDE: same expressions as train.py:546-557 on the recorded G-phase prediction of each step
    ade = fde = 0.0
    shapes = []
    calls_per_step = 2 * (m.n_unrolling_steps + 1) + 1
    preds_per_step = (m.n_unrolling_steps + 1) + 1
    for s in range(n_steps):
        pred_hat = rec.predicts[s * preds_per_step + preds_per_step - 1]
        real_pred_4d = rec.d_calls[s * calls_per_step + 1][1]
        pred = real_pred_4d[:, :, :2]
        err_all = torch.pow((pred_hat[:, :, :2] - pred) / m.ss, 2).sum(dim=2).sqrt()
        ade += err_all.sum().item() / m.n_next
        fde += err_all[:, -1].sum().item()
        shapes.append(pred_hat.shape[0])
    ade /= m.n_train_samples
    fde /= m.n_train_samples
    printed = buf.getvalue()
    out = dict(losses=losses, ade=ade, fde=fde, step_agents=np.asarray(shapes),
               uniform=np.asarray(rec.uniform, np.float64).reshape(n_steps, 2),
               printed=np.array(printed), ss=np.float64(m.ss), threads=torch.get_num_threads(),
               n_train_samples=m.n_train_samples, train_size=m.train_size, batch_size=batch_size,
               use_social=use_social, seed=seed)
    for s in range(n_steps):
        out["noise.%d" % s] = rec.noise[s].numpy()
    out.update(w0)
    out.update(w1)
    return m, rec, out


def run_one_step(dataset, use_social, seed=0):
    """batch_size = all training agents -> train() runs exactly one packed step."""
    batches = np.asarray(dataset["batches"])
    train_size = max(1, (len(batches) * 4) // 5)
    B = int(batches[train_size - 1][1])
    m, rec, out = run_epoch(dataset, B, use_social, seed)
    assert len(rec.g_grads) == 1, len(rec.g_grads)
    out["hT"] = rec.dec_first[-1][0].numpy()
    out["S"] = rec.dec_first[-1][1].numpy()
    out["pred_hat_4d"] = rec.predicts[-1].numpy()
    out["dpred_hat_4d"] = rec.dpred[-1].numpy()
    names = ["d0_fake", "d0_real", "d1_fake", "d1_real", "g_fake"]
    for nm, (obsv4, pred4, label, code) in zip(names, rec.d_calls):
        out[nm + ".label"] = label.numpy()
        out[nm + ".code"] = code.numpy()
    out["obsv_4d"] = rec.d_calls[0][0].numpy()
    out["pred_4d"] = rec.d_calls[1][1].numpy()
    for u, g in enumerate(rec.d_grads):
        for k, v in g.items():
            out["dgrad%d.%s" % (u, k)] = v
    for k, v in rec.g_grads[0].items():
        out["ggrad." + k] = v
    m._cleanup()
    return out


def social_ops_case():
    sizes = [3, 1, 8, 2, 17, 5]
    ds = sw_oracle.synth_dataset(len(sizes), sizes, seed=7)
    m = import_reference(ds, 64, seed=3, use_social=True)
    B = int(np.sum(sizes))
    obsv = m.dataset_obsv[:B]
    sb = np.asarray(ds["batches"])
    obsv_4d = m.get_traj_4d(obsv, [])
    torch.manual_seed(11)
    h = torch.randn(B, 64) * 0.5
    with torch.no_grad():
        feats = m.SocialFeatures(obsv_4d, sb)
        emb = m.feature_embedder(feats, sb)
        S = m.attention(emb, h, sb)
        last = obsv_4d[:, -1]
        pairs = [(0, 1), (2, 0), (4, 9), (11, 4), (14, 30), (30, 14), (31, 35)]
        dca = [m.DCA(last[i], last[j]).item() for i, j in pairs]
        bear = [m.Bearing(last[i], last[j]).item() for i, j in pairs]
    out = dict(obsv=obsv.numpy(), batches=sb, h=h.numpy(), features=feats.numpy(), S=S.numpy(),
               pairs=np.asarray(pairs), dca_scalar=np.asarray(dca, np.float32),
               bearing_scalar=np.asarray(bear, np.float32), threads=torch.get_num_threads())
    # embeddings only on the block diagonal (the dense tensor is B*B*64)
    for s, (a, b) in enumerate(sb):
        out["emb.%d" % s] = emb[a:b, a:b].numpy()
    out.update(flat_state(m, "w0."))
    m._cleanup()
    return out


def test_eval_case():
    ds = sw_oracle.synth_dataset(10, [4, 6, 3, 8, 5, 7, 2, 8, 5, 3], seed=21)
    m = import_reference(ds, 64, seed=5, use_social=True)
    rec = Recorder(m)
    wr = os.path.join(m._root, "preds")
    os.makedirs(wr)
    torch.manual_seed(123)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        m.test(n_gen_samples=4, write_to_file=wr)
    rec.close()
    K = 4
    out = dict(printed=np.array(buf.getvalue()), threads=torch.get_num_threads(), ss=np.float64(m.ss),
               n_test_samples=m.n_test_samples, train_size=m.train_size)
    out.update(flat_state(m, "w0."))
    # metrics: same expressions as train.py:587, 602-614
    ade_avg = fde_avg = ade_min = fde_min = 0.0
    for si, batch_i in enumerate(m.test_batches):
        pred = m.dataset_pred[batch_i[0]:batch_i[1]]
        errs = []
        for kk in range(K):
            ph = rec.predicts[si * K + kk]
            out["pred_hat.%d.%d" % (si, kk)] = ph.numpy()
            out["noise.%d.%d" % (si, kk)] = rec.noise[si * K + kk].numpy()
            errs.append(torch.pow((ph[:, :, :2] - pred) / m.ss, 2).sum(dim=2, keepdim=True).sqrt().unsqueeze(0))
        errs = torch.cat(errs)
        fde_min += errs[:, :, -1].min(0, keepdim=True)[0].sum().item()
        ade_min += errs.mean(2).min(0, keepdim=True)[0].sum().item()
        fde_avg += errs[:, :, -1].mean(0, keepdim=True).sum().item()
        ade_avg += errs.mean(2).mean(0, keepdim=True).sum().item()
    n = m.n_test_samples
    out["metrics"] = np.asarray([ade_avg / n, fde_avg / n, ade_min / n, fde_min / n])
    for f in sorted(glob.glob(os.path.join(wr, "*.npz"))):
        z = np.load(f)
        tag = os.path.basename(f)[:-4]
        for k in z.files:
            out["npz.%s.%s" % (tag, k)] = z[k]
    out["npz_files"] = np.array(sorted(os.path.basename(f) for f in glob.glob(os.path.join(wr, "*.npz"))))
    for k in ("obsvs", "preds", "times", "batches"):
        out["ds." + k] = ds[k]
    m._cleanup()
    return out


def multi_epoch_case(dataset, n_epochs=5, batch_size=64, seed=0):
    """n_epochs of the reference's train() back to back (the module-level epoch loop of train.py:646-650 without
    the checkpointing), recording every draw so that the runs under test can be fed identically."""
    m = import_reference(dataset, batch_size, seed, True)
    out = dict(threads=torch.get_num_threads(), batch_size=batch_size, n_epochs=n_epochs)
    out.update(flat_state(m, "w0."))
    rec = Recorder(m)
    np.random.seed(seed)
    per = 3 * (m.n_unrolling_steps + 1) + 3
    calls_per_step = 2 * (m.n_unrolling_steps + 1) + 1
    preds_per_step = (m.n_unrolling_steps + 1) + 1
    s_done = 0
    for ep in range(n_epochs):
        m.epoch = ep + 1
        with contextlib.redirect_stdout(io.StringIO()):
            m.train()
        n_steps = len(rec.g_grads) - s_done
        ade = fde = 0.0
        for s in range(s_done, s_done + n_steps):       # same expressions as train.py:546-557
            pred_hat = rec.predicts[s * preds_per_step + preds_per_step - 1]
            pred = rec.d_calls[s * calls_per_step + 1][1][:, :, :2]
            err = torch.pow((pred_hat[:, :, :2] - pred) / m.ss, 2).sum(dim=2).sqrt()
            ade += err.sum().item() / m.n_next
            fde += err[:, -1].sum().item()
        out["ade.%d" % ep], out["fde.%d" % ep] = ade / m.n_train_samples, fde / m.n_train_samples
        out["losses.%d" % ep] = np.asarray(rec.mse[s_done * per:(s_done + n_steps) * per], np.float64).reshape(n_steps, per)
        out["uniform.%d" % ep] = np.asarray(rec.uniform[2 * s_done:2 * (s_done + n_steps)], np.float64).reshape(n_steps, 2)
        for s in range(n_steps):
            out["noise.%d.%d" % (ep, s)] = rec.noise[s_done + s].numpy()
        s_done += n_steps
    rec.close()
    m._cleanup()
    return out


def ref_checkpoint_case(dataset, batch_size=64, seed=0):
    """A checkpoint the UNMODIFIED reference writes itself, and the reference resuming from it.
      1. train.py runs as it ships (use_social hard-coded False, train.py:83) for 50 epochs; at epoch 50 its own
         torch.save (train.py:651-663) writes ../trained_models/socialWays-hotel.pt.  Stored: the 5 state_dicts and
         BOTH Adam state dicts exactly as found in that file - the generator optimizer holds state for parameter
         indices 8..21 only (attention / feature_embedder never receive gradients without the social block).
      2. a second import finds that file, so train.py:622-634 loads it (start_epoch 51); one more train() call from
         there is recorded: the 9 MSE terms per step, every RNG draw, ADE/FDE = the resumed epoch 51."""
    m = import_reference(dataset, batch_size, seed, False, epochs=50)
    pt = os.path.join(m._root, "trained_models", "socialWays-hotel.pt")
    assert os.path.isfile(pt), "the reference did not write its checkpoint"
    ck = torch.load(pt, weights_only=False)
    out = dict(threads=torch.get_num_threads(), batch_size=batch_size, epoch=int(ck["epoch"]))
    for name in ("attentioner_dict", "feature_embedder_dict", "encoder_dict", "decoder_dict", "D_dict"):
        for k, v in ck[name].items():
            out["ck.%s.%s" % (name, k)] = v.detach().numpy()
    for name in ("pred_optimizer", "D_optimizer"):
        sd = ck[name]
        grp = sd["param_groups"][0]
        out["ck.%s.params" % name] = np.asarray(grp["params"], np.int64)
        out["ck.%s.hyper" % name] = np.asarray([grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"]],
                                               np.float64)
        out["ck.%s.present" % name] = np.asarray(sorted(sd["state"].keys()), np.int64)
        for i, st in sd["state"].items():
            out["ck.%s.%d.step" % (name, i)] = np.float64(float(st["step"]))
            out["ck.%s.%d.exp_avg" % (name, i)] = st["exp_avg"].numpy()
            out["ck.%s.%d.exp_avg_sq" % (name, i)] = st["exp_avg_sq"].numpy()
    tmp = tempfile.mkdtemp(prefix="swck_")
    keep = os.path.join(tmp, "ck.pt")
    shutil.copy(pt, keep)
    m._cleanup()
    # --- resume: a differently seeded process (its fresh init must be overwritten by the file)
    m2 = import_reference(dataset, batch_size, seed + 123, False, epochs=50, checkpoint_from=keep)
    assert m2.start_epoch == 51
    m2.epoch = 51
    rec = Recorder(m2)
    torch.manual_seed(seed + 7)
    np.random.seed(seed + 7)
    with contextlib.redirect_stdout(io.StringIO()):
        m2.train()
    rec.close()
    n_steps = len(rec.g_grads)
    per = 3 * (m2.n_unrolling_steps + 1) + 3
    calls_per_step = 2 * (m2.n_unrolling_steps + 1) + 1
    preds_per_step = (m2.n_unrolling_steps + 1) + 1
    ade = fde = 0.0
    for s in range(n_steps):
        pred_hat = rec.predicts[s * preds_per_step + preds_per_step - 1]
        pred = rec.d_calls[s * calls_per_step + 1][1][:, :, :2]
        err = torch.pow((pred_hat[:, :, :2] - pred) / m2.ss, 2).sum(dim=2).sqrt()
        ade += err.sum().item() / m2.n_next
        fde += err[:, -1].sum().item()
    out["resume.ade"], out["resume.fde"] = ade / m2.n_train_samples, fde / m2.n_train_samples
    out["resume.losses"] = np.asarray(rec.mse, np.float64).reshape(n_steps, per)
    out["resume.uniform"] = np.asarray(rec.uniform, np.float64).reshape(n_steps, 2)
    for s in range(n_steps):
        out["resume.noise.%d" % s] = rec.noise[s].numpy()
    out.update(flat_state(m2, "resume.w1."))
    m2._cleanup()
    shutil.rmtree(tmp, ignore_errors=True)
    return out


VARIANTS = {
    "l2": dict(use_l2_loss=True),
    "variety": dict(use_variety_loss=True),
    "unroll0": dict(n_unrolling_steps=0),
    "unroll2": dict(n_unrolling_steps=2),
    "noinfo": dict(use_info_loss=False),
}


def variants_case(dataset):
    """One packed step of `dataset` (social on, seed 0: same initial weights and RNG draws as
    syn_s16a8_on) per switch setting."""
    batches = np.asarray(dataset["batches"])
    train_size = max(1, (len(batches) * 4) // 5)
    B = int(batches[train_size - 1][1])
    out = dict(names=np.array(sorted(VARIANTS)), threads=torch.get_num_threads())
    for name, ov in sorted(VARIANTS.items()):
        m, rec, o = run_epoch(dataset, B, True, 0, overrides=ov)
        assert len(rec.g_grads) == 1
        if "w0.D.lstm.weight_ih_l0" not in out:
            out.update({k: v for k, v in o.items() if k.startswith("w0.")})
            out["noise"], out["uniform"], out["ss"] = o["noise.0"], o["uniform"], o["ss"]
        else:
            assert all(np.array_equal(out[k], v) for k, v in o.items() if k.startswith("w0."))
            assert np.array_equal(out["noise"], o["noise.0"]) and np.array_equal(out["uniform"], o["uniform"])
        out[name + ".losses"] = o["losses"][0]
        out[name + ".ade_fde"] = np.asarray([o["ade"], o["fde"]])
        out[name + ".n_d_updates"] = len(rec.d_grads)
        for k, v in rec.d_grads[-1].items():
            out["%s.dgrad_last.%s" % (name, k)] = v
        for k, v in rec.g_grads[0].items():
            out["%s.ggrad.%s" % (name, k)] = v
        for k, v in o.items():
            if k.startswith("w1.D."):
                out[name + "." + k] = v
        m._cleanup()
    return out


def reference_stat_functions():
    """compute_1nn / compute_wasserstein from /root/reference/calc_statistics.py without running the script
    body (it loads ../data/toy/toy-768.npz and plots at import time)."""
    import ast
    path = os.path.join(REF, "calc_statistics.py")
    tree = ast.parse(open(path).read(), path)
    keep = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) or
            (isinstance(n, ast.FunctionDef) and n.name in ("compute_1nn", "compute_wasserstein"))]
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["compute_1nn"], ns["compute_wasserstein"]


def toy_stats_case(toy6):
    """real_samples as calc_statistics.py:167-176 builds them (obs+pred concatenated, (-1,6,4,2)[:20]);
    fakes = the same conditions with K different perturbations of the futures (what preds_our holds)."""
    one_nn, emd = reference_stat_functions()
    real = np.concatenate((toy6["obsvs"], toy6["preds"]), axis=1).reshape((-1, 6, 4, 2))[:20]
    rng = np.random.default_rng(5)
    out = dict(real=real.astype(np.float32), sigmas=np.asarray([0.0, 0.01, 0.05, 0.3]))
    for i, sg in enumerate(out["sigmas"]):
        fake = real.copy()
        perm = rng.permutation(20)                      # sample order of a generator is arbitrary
        fake[:, :, 2:] = real[perm][:, :, 2:] + rng.normal(0, sg, size=real[:, :, 2:].shape).astype(np.float32)
        fake = fake.astype(np.float32)
        out["fake.%d" % i] = fake
        out["one_nn.%d" % i] = np.asarray(one_nn(out["real"], fake), np.float64)
        out["emd.%d" % i] = np.float64(emd(out["real"], fake))
    # a ragged case: fewer pedestrians / longer tracks / obsv_len 3 (the functions are shape-generic)
    a = rng.normal(0, 1, size=(7, 3, 9, 2)).astype(np.float32)
    b = (a[rng.permutation(7)] + rng.normal(0, 0.2, size=a.shape)).astype(np.float32)
    out["g.real"], out["g.fake"] = a, b
    out["g.one_nn"] = np.asarray(one_nn(a, b, 3), np.float64)
    out["g.emd"] = np.float64(emd(a, b, 3))
    return out


def biwi_case():
    """Inputs from socialways_amd.data.synth_crowd_frames (pure numpy), expected outputs from the
    reference's own parser and window extraction."""
    sys.path.insert(0, os.path.dirname(HERE))
    from socialways_amd import data as D
    fr, ids, pos, vel = D.synth_crowd_frames()
    d = tempfile.mkdtemp(prefix="swbiwi_")
    path = os.path.join(d, "obsmat.txt")
    D.write_biwi_obsmat(path, fr, ids, pos, vel)
    old_path = list(sys.path)
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    try:
        from utils.parse_utils import BIWIParser, create_dataset
        p = BIWIParser()
        p.load(path)
        ob, pr, ti, ba = create_dataset(p.p_data, p.t_data, range(p.t_data[0][0], p.t_data[-1][-1], p.interval), 8, 12)
    finally:
        sys.path[:] = old_path
        shutil.rmtree(d, ignore_errors=True)
    return dict(frames=fr, ids=ids, pos=pos, vel=vel, obsvs=ob, preds=pr, times=np.asarray(ti),
                batches=ba.astype(np.int64), interval=p.interval)


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-18s %7.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024, len(d)))


def main():
    t0 = time.time()
    torch.set_num_threads(8)
    only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else None      # e.g. `make_golden.py syn_variants,biwi_synth`
    syn = sw_oracle.synth_dataset(20, 8, seed=1234)                  # 16 train scenes x 8
    if only is None or "syn_variants" in only:
        save("syn_variants", variants_case(syn))
    if only is None or "syn_big_on" in only:      # scenes above 64 agents (the build's row-block kernels) straight from the reference
        big = sw_oracle.synth_dataset(5, [70, 3, 90, 8, 2], seed=77)
        out = run_one_step(big, True)
        for k in ("obsvs", "preds", "batches"):
            out["ds." + k] = big[k]
        save("syn_big_on", out)
    if only is None or "toy_multi" in only:
        save("toy_multi", multi_epoch_case(toy_dataset(768, 8, 3)))
    if only is None or "ref_checkpoint" in only:
        save("ref_checkpoint", ref_checkpoint_case(toy_dataset(768, 8, 3)))
    if only is None or "biwi_synth" in only:
        save("biwi_synth", biwi_case())
    if only is None or "toy_stats" in only:
        save("toy_stats", toy_stats_case(toy_dataset(768, 6, 3)))
    if only is not None:
        print("done in %.1fs" % (time.time() - t0))
        return
    toy8 = toy_dataset(768, 8, 3)
    toy6 = toy_dataset(768, 6, 3)
    save("toy_768_8_3", toy8)
    save("toy_768_6_3", toy6)
    for flag, tag in ((False, "off"), (True, "on")):
        m, rec, out = run_epoch(toy8, 64, flag)
        m._cleanup()
        print("  toy epoch social=%s  ADE/FDE=%.6f/%.6f  printed=%s" % (flag, out["ade"], out["fde"], str(out["printed"]).strip()))
        save("toy_b64_" + tag, out)
    for flag, tag in ((False, "off"), (True, "on")):
        out = run_one_step(syn, flag)
        for k in ("obsvs", "preds", "batches"):
            out["ds." + k] = syn[k]
        save("syn_s16a8_" + tag, out)
    sizes = [1, 2, 5, 8, 3, 16, 1, 7, 33, 64, 2, 1, 4]              # last 3 scenes = held-out 1/5
    rag = sw_oracle.synth_dataset(len(sizes), sizes, seed=99)
    out = run_one_step(rag, True)
    for k in ("obsvs", "preds", "batches"):
        out["ds." + k] = rag[k]
    save("syn_ragged_on", out)
    save("social_ops", social_ops_case())
    save("test_eval", test_eval_case())
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
