#!/usr/bin/env python
"""train_joint.py counterpart (reference train_joint.py:191-291) on synthetic scans: Adam, step LR decay,
batch of 3 scenes, masked MSE(xyz) + MSE(log scale) + CE(class); one process per GPU with DDP when launched
through torch.distributed.run (the reference is single-GPU).

    python scripts/train_joint.py [--epochs 2] [--scenes 6] [--points 20000] [--batch 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import dist as cvd  # noqa: E402
from canonicalvoting_amd import train  # noqa: E402
from canonicalvoting_amd.data import (ScanNetXYZProbMultiDataset, SyntheticScanDataset, collate_fn,  # noqa: E402
                                     load_config)
from canonicalvoting_amd.minkunet import MinkUNet34C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--scenes", type=int, default=6)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--save", default=None)
    ap.add_argument("--config", default=None, help="the reference's config.yaml: train on real ScanNet/Scan2CAD files")
    a = ap.parse_args()
    cfg = load_config(a.config, category="all") if a.config else None
    world, rank, local = cvd.world()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cvd.init("nccl", dev)
    torch.manual_seed(0)
    model = MinkUNet34C(6 if (cfg and cfg.use_xyz) else 3, 6 * 9 + 9 + 1).to(dev)
    net = train.make_ddp(model, dev) if world > 1 else model
    base_lr, wd, steps, rates = a.lr, 0.0, (80, 120, 160), (0.1, 0.1, 0.1)
    bn_step, bn_rate = 20.0, 0.5                                               # config/config.yaml opt.bn_decay_*
    first_epoch, last_epoch = 0, a.epochs - 1
    if cfg:
        # train_joint.py:204-206,219-223,235: the optimizer and the schedule come from the config
        base_lr, wd = float(cfg.opt.learning_rate), float(cfg.weight_decay)
        steps = [int(x) for x in str(cfg.opt.lr_decay_steps).split(",")]
        rates = [float(x) for x in str(cfg.opt.lr_decay_rates).split(",")]
        first_epoch, last_epoch = int(cfg.start_epoch), int(cfg.max_epoch)      # range(start_epoch, max_epoch + 1)
        bn_step, bn_rate = float(cfg.opt.bn_decay_step), float(cfg.opt.bn_decay_rate)
    opt = train.make_optimizer(model, lr=base_lr, weight_decay=wd)
    if cfg:
        # train_joint.py:205-211; every rank reads its own slice of the scans (DistributedSampler)
        ds = ScanNetXYZProbMultiDataset(cfg, training=True, augment=cfg.augment)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=sampler is None, sampler=sampler,
                                             collate_fn=collate_fn, drop_last=True, num_workers=cfg.num_workers)
    else:
        ds = SyntheticScanDataset(a.scenes, a.points, seed0=1000 * rank)      # every rank owns its scenes
        loader = torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=True, collate_fn=collate_fn, drop_last=True)
    for epoch in range(first_epoch, last_epoch + 1):
        train.adjust_learning_rate(opt, epoch, base_lr, steps, rates)
        train.set_bn_momentum(model, train.bn_momentum(epoch, bn_step, bn_rate))   # train_joint.py:224-225,239
        net.train()
        t0, tot, n = time.perf_counter(), 0.0, 0
        for _, coords, feats, xyz, scale, cls in loader:
            feats = feats.to(dev)
            feats[:, -3:] = feats[:, -3:] * 2.0 - 1.0                         # train_joint.py:248-249: colour columns only
            loss, _ = train.train_step(net, opt, coords.to(dev), feats, xyz.to(dev), scale.to(dev), cls.to(dev))
            tot += float(loss)
            n += 1
        if rank == 0:
            print("epoch %d  loss %.4f  %.1f s" % (epoch, tot / max(n, 1), time.perf_counter() - t0), flush=True)
        if a.save and rank == 0 and (epoch % 10 == 0 or epoch == last_epoch):
            # train_joint.py:290-291: one file per saved epoch, 'epoch{N}.pth' (here under the --save prefix)
            stem, ext = os.path.splitext(a.save)
            torch.save(model.state_dict(), "%s_epoch%d%s" % (stem, epoch, ext or ".pth"))
    cvd.finalize()


if __name__ == "__main__":
    main()
