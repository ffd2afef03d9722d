"""End-to-end example on the reference's toy data (create_toy.py semantics): what a user of crowdbotp/socialways'
`train.py` runs, on the MI355X path.

    python examples/train_toy.py --epochs 20 --batch-size 64 --out /tmp/sw_toy

* toy tracks (2 observed + 2 future points, 6 conditions x 3 modes)         socialways_amd.toy_tracks
* normalisation + 4/5 train split by scene (train.py:89-127)                 socialways_amd.SceneDataset
* epochs of train() with the reference's prints (train.py:559-560)          SocialWaysTrainer.train_epoch
* test() every 5 epochs: avg / min-over-K ADE, FDE + prediction npz files    SocialWaysTrainer.test
* checkpoint in the reference's 8-key format (train.py:651-663)             SocialWaysTrainer.save
* the toy statistics of calc_statistics.py on the written predictions        socialways_amd.stats
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import socialways_amd as sw  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=64)            # train.py --batch-size
    ap.add_argument("--unrolling-steps", type=int, default=1)        # train.py --unrolling-steps
    ap.add_argument("--hidden-size", type=int, default=64)           # train.py --hidden-size (8, 16, .. 64)
    ap.add_argument("--social", type=int, default=1)
    ap.add_argument("--n-samples", type=int, default=768)
    ap.add_argument("--test-every", type=int, default=5)             # train.py:665
    ap.add_argument("--k", type=int, default=20, help="samples per scene in test()")
    ap.add_argument("--out", default="/tmp/sw_toy")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    toy = sw.toy_tracks(args.n_samples, n_conditions=6, n_modes=3)
    data = sw.SceneDataset(toy["obsvs"], toy["preds"], toy["batches"], toy["times"], device="cuda:0")
    tr = sw.SocialWaysTrainer(data.n_next, hidden_size=args.hidden_size, use_social=bool(args.social),
                              n_unrolling_steps=args.unrolling_steps,
                              device="cuda:0")
    real = np.concatenate((toy["obsvs"], toy["preds"]), axis=1).reshape((-1, 6, 4, 2))[:args.k]
    pred_root = os.path.join(args.out, "preds")
    os.makedirs(args.out, exist_ok=True)
    for epoch in range(1, args.epochs + 1):
        t0 = time.perf_counter()
        ade, fde, losses, _ = tr.train_epoch(data, args.batch_size)
        print("Epc=%4d, Train ADE,FDE = (%.3f, %.3f) | time = %.1f  | D/G/Info losses %.4f %.4f %.4f"
              % (epoch, ade, fde, time.perf_counter() - t0, losses[:, 0].mean() + losses[:, 2].mean(),
                 losses[:, -2].mean(), losses[:, -1].mean()))
        if epoch % args.test_every == 0 or epoch == args.epochs:
            m = tr.test(data, n_gen_samples=args.k, write_to_file=os.path.join(pred_root, str(epoch)))
            print("Avg ADE,FDE = (%.3f, %.3f) | Min(%d) ADE,FDE = (%.3f, %.3f)" % (m[0], m[1], args.k, m[2], m[3]))
            tr.save(os.path.join(args.out, "toy.pt"))
    s1, sw_ = sw.stats.calc_and_store_stats(pred_root, real, 2, 2, stats_file=os.path.join(args.out, "stats%d.npz" % args.k))
    for ep in sorted(s1):
        print("epoch = %d, EMD = %.5f, 1nn = %.5f" % (ep, sw_[ep], s1[ep]))
    return tr, s1, sw_


if __name__ == "__main__":
    main()
