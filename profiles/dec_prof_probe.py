import torch, bench
from canonicalvoting_amd import decode
from canonicalvoting_amd.hough import HoughVoting
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
hv = HoughVoting(bench.RES, bench.NUM_ROTS)
for npts, large in ((80000, False), (300000, True)):
    s = bench.ResidentScene(0, npts, dev, large)
    g = hv(s.points, s.xyz, s.scale, s.prob)
    for _ in range(2):
        raw = decode.decode_boxes(g[0], g[1], g[2], s.points, s.xyz, s.prob, s.cls, bench.RES)
    torch.cuda.synchronize()
    print(npts, "candidates", len(raw["cand_idx"]), flush=True)
