"""One fine-level masked 3x3x3 conv (hl operands, 3 mask groups as the network plans them) repeated, conv_hd (8 waves x 2
stages) against conv_hh (half-chunk stages, two workgroups per CU): per-call time (conv + finish launch).
argv: repeats [rows = 80000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
sc = make_scene(3, n)
c4 = torch.cat([torch.zeros((n, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]
for ts in (1, 2):
    nbr = cm.kernel_map(3, ts)
    N = nbr.shape[0]
    perms = cm.mask_perms(3, ts, 3)
    for cin, cout in ((96, 96), (128, 96), (64, 64), (32, 32)):
        x = ME.to_hl(torch.randn(N, cin, device=dev))
        w = torch.randn(27, cin, cout, device=dev) * 0.02
        line = "ts%d rows %6d %3d->%2d:" % (ts, N, cin, cout)
        ys = []
        for shape in (2, 3):
            prev = ME.set_option("hd_shape", shape)
            pm = ME.set_option("hd_mask", 7)
            pr = ME.set_option("hd_min_rows", 1)
            try:
                for _ in range(3):
                    y = ME.conv_forward_masked(x, w, nbr, perms, N, relu=True, pieces=2, in_hl=True, out_hl=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    y = ME.conv_forward_masked(x, w, nbr, perms, N, relu=True, pieces=2, in_hl=True, out_hl=True)
                e1.record()
                torch.cuda.synchronize()
            finally:
                ME.set_option("hd_shape", prev); ME.set_option("hd_mask", pm); ME.set_option("hd_min_rows", pr)
            ys.append(y)
            line += "  shape %d %.1f us" % (shape, e0.elapsed_time(e1) / reps * 1e3)
        print(line, " identical" if torch.equal(ys[0], ys[1]) else " DIFFERENT", flush=True)
