"""Mask-sorted 3x3x3 convs of the two fine levels with 1 / 2 / 4 offset groups (hl-format in and out, fp16 pairs):
conv + finish time per call.  One group = one row order sorted by the full 27-bit mask: 2.25 x the MFMA blocks of four
groups (profiles: live (32-row block, offset) fraction 0.52 vs 0.23) but no partial sums and no finish kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
N = 80000
sc = make_scene(3, N)
c4 = torch.cat([torch.zeros((N, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]
REPS = 20
for ts in (1, 2):
    n = cm.num_rows(ts)
    nbr = cm.kernel_map(3, ts)
    for cin, cout in ((96, 96), (128, 96), (32, 32)):
        if ts == 1 and cin == 32:
            continue
        x = ME.to_hl(torch.randn(n, cin, device=dev))
        w = torch.randn(27, cin, cout, device=dev) * 0.02
        sc_ = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
        res = ME.to_hl(torch.randn(n, cout, device=dev))
        out = torch.empty(n, cout, device=dev)
        line = "ts%d %6d rows %3d->%3d:" % (ts, n, cin, cout)
        ref = None
        for G in (4, 2, 1):
            perms = cm.mask_perms(3, ts, G)
            kw = dict(scale=sc_, shift=sh, residual=res, relu=True, out=out, in_hl=True, out_hl=True, res_hl=True, pieces=2)
            if G > 1:
                f = lambda: ME.conv_forward(x, w, nbr, n, row_perm=perms, perm_groups=G, **kw)
            else:
                f = lambda: ME.conv_forward(x, w, nbr, n, row_perm=perms[0].contiguous(), **kw)
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                f()
            e1.record()
            torch.cuda.synchronize()
            y = ME.from_hl(out)
            if ref is None:
                ref = y
            err = float((y - ref).abs().max() / ref.abs().max())
            line += "  G=%d %7.1f us (err %.1e)" % (G, e0.elapsed_time(e1) / REPS * 1e3, err)
        print(line, flush=True)
