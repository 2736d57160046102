"""a few training steps at 3 x 80k: does the range flag stay down with the hl twins of the gradients?  per-layer maxima of |dx|"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from canonicalvoting_amd import me as ME, train
from canonicalvoting_amd.minkunet import MinkUNet34C
dev = torch.device('cuda')
batch = bench.train_batch(0, 3, 80000, dev)
torch.manual_seed(0)
model = MinkUNet34C(3, 64).cuda().train()
opt = train.make_optimizer(model)
for i in range(5):
    loss, _ = train.train_step(model, opt, *batch)
    torch.cuda.synchronize()
    sl = model.__dict__.get("_grad_slots")
    cur = sl.buf[:len(sl.index), 0:4096].view(torch.float32).max(1).values if sl is not None else None
    print("step %d loss %.4f flag %d fallbacks %d hl dgrads %d" % (i, float(loss), int(ME.range_flag(dev)[0]), getattr(model, "train_range_fallbacks", 0), ME.TRAIN_COUNTERS["hl_dgrad"]),
          "max|dx| per layer: min %.2e max %.2e" % (float(cur.min()), float(cur.max())) if cur is not None else "")
