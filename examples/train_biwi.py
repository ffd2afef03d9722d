"""End-to-end example on a BIWI / ETH-format recording: the on-disk path a user of crowdbotp/socialways takes with
`create_dataset.py` + `train.py`, on the MI355X path.

    python examples/train_biwi.py --obsmat /data/eth/hotel/obsmat.txt --epochs 50 --out /tmp/sw_hotel
    python examples/train_biwi.py --epochs 10          # no recording at hand: a synthetic crowd in the same file format

* obsmat.txt [frame id px pz py vx vz vy] -> 8 + 12 step windows, one scene per timestamp   socialways_amd.biwi_to_npz
                                                               (utils/parse_utils.py:231-320, :457-508; create_dataset.py)
* the npz train.py loads ('../hotel-8-12.npz', train.py:56, 89-127)                          SceneDataset.from_npz
* packed batches of whole scenes up to --batch-size agents (train.py:446-456), train(), test() every 5 epochs with the
  prediction npz files visualize.py / calc_statistics.py read, checkpoint in the reference's format
* data parallel: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_biwi.py ...`
  (one rank per GPU over RCCL; every packed batch is sharded scene-aligned, rank 0 evaluates and saves)
  `SW_ALLREDUCE=direct` exchanges the gradients through the library's own two-hop kernel over hipIpc-mapped peer buffers
  (one launch per optimizer step: exchange + Adam, inside the step's hipGraph) instead of RCCL's ring
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
    ap.add_argument("--obsmat", default=None, help="BIWI obsmat.txt; default: a synthetic recording written to --out")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=256)           # train.py --batch-size (agents per packed batch)
    ap.add_argument("--hidden-size", type=int, default=64)
    ap.add_argument("--social", type=int, default=1)
    ap.add_argument("--test-every", type=int, default=5)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--out", default="/tmp/sw_biwi")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    pg = None
    if world > 1:                                                    # launched by torch.distributed.run: one rank per GPU
        torch.cuda.set_device(dev)
        torch.distributed.init_process_group("nccl", device_id=dev)
        pg = torch.distributed.group.WORLD
    os.makedirs(args.out, exist_ok=True)
    obsmat = args.obsmat
    if obsmat is None:
        obsmat = os.path.join(args.out, "obsmat.txt")
        if rank == 0:
            fr, ids, pos, vel = sw.synth_crowd_frames(n_frames=400, n_ped=160, interval=6, seed=args.seed + 3)
            sw.write_biwi_obsmat(obsmat, fr, ids, pos, vel)
    npz = os.path.join(args.out, "crowd-8-12.npz")
    if rank == 0:
        obsvs, preds, times, batches = sw.biwi_to_npz(obsmat, npz, 8, 12)
        print("%s: %d samples in %d scenes (largest %d agents)" % (obsmat, len(obsvs), len(batches),
                                                                    int(np.max(batches[:, 1] - batches[:, 0]))))
    if world > 1:
        torch.distributed.barrier()
    data = sw.SceneDataset.from_npz(npz, device=dev)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    tr = sw.SocialWaysTrainer(data.n_next, hidden_size=args.hidden_size, use_social=bool(args.social), device=dev,
                              process_group=pg)
    for epoch in range(1, args.epochs + 1):
        t0 = time.perf_counter()
        ade, fde, losses, sizes = tr.train_epoch(data, args.batch_size)
        if rank == 0:
            print("Epc=%4d, Train ADE,FDE = (%.3f, %.3f) | time = %.2f | %d packed batches | D/G losses %.4f %.4f"
                  % (epoch, ade, fde, time.perf_counter() - t0, len(sizes), losses[:, 0].mean() + losses[:, 2].mean(),
                     losses[:, -2].mean()))
        if rank == 0 and (epoch % args.test_every == 0 or epoch == args.epochs):
            m = tr.test(data, n_gen_samples=args.k, write_to_file=os.path.join(args.out, "preds", str(epoch)))
            print("Avg ADE,FDE = (%.3f, %.3f) | Min(%d) ADE,FDE = (%.3f, %.3f)" % (m[0], m[1], args.k, m[2], m[3]))
            tr.save(os.path.join(args.out, "socialWays-crowd.pt"), epoch=epoch)
    if world > 1:
        tr.close()              # captured collectives and the direct exchange's buffers go before their process group
        torch.distributed.destroy_process_group()
    return tr


if __name__ == "__main__":
    main()
