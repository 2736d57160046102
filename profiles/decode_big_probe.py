"""300k-point scenes: cells listed for the greedy walk, candidates examined, time of the decode stage.
PYTHONPATH=. python profiles/decode_big_probe.py"""
import numpy as np, torch
import bench
from canonicalvoting_amd import decode, hv_cuda
from canonicalvoting_amd.hough import HoughVoting

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
hv = HoughVoting(bench.RES, bench.NUM_ROTS)
for seed in range(3):
    s = bench.ResidentScene(seed, 300000, dev, True)
    g_obj, g_rot, g_scale = hv(s.points, s.xyz, s.scale, s.prob)
    listed = int((g_obj >= decode.thresh_high).sum())
    for _ in range(3):
        raw = decode.decode_boxes(g_obj, g_rot, g_scale, s.points, s.xyz, s.prob, s.cls, bench.RES)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        raw = decode.decode_boxes(g_obj, g_rot, g_scale, s.points, s.xyz, s.prob, s.cls, bench.RES)
    e1.record()
    torch.cuda.synchronize()
    print("scene %d: grid %s, listed cells %d, candidates %d, boxes %d, decode %.3f ms" %
          (seed, tuple(g_obj.shape), listed, len(raw["cand_idx"]), len(raw["boxes"]), e0.elapsed_time(e1) / 5))
