"""decode on the teacher predictions of bench scene 0, timed with HIP events (whole op) - run under rocprofv3 for the
per-kernel split (profiles/decode_prof.sh)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import decode, hv_cuda
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions
dev = torch.device("cuda:0")
sc = make_scene(0, n_points=80000)
xyz, scale, prob, cls = synth_predictions(sc)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p, x, s, o, c = t(sc.points), t(xyz), t(scale), t(prob), t(cls)
hv = HoughVoting(0.03, 120)
with torch.no_grad():
    g = hv(p, x, s, o)
kw = dict(allow_truncation=True)
for _ in range(3):
    raw = decode.decode_boxes(*g, p, x, o, c, 0.03, **kw)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
t0 = time.perf_counter()
for i in range(20):
    ev[i].record()
    raw = decode.decode_boxes(*g, p, x, o, c, 0.03, **kw)
ev[20].record()
torch.cuda.synchronize()
print("dbg 0 cands", len(raw["cand_idx"]), "boxes", len(raw["boxes"]),
      "event ms/decode %.4f" % (ev[0].elapsed_time(ev[20]) / 20), "wall ms/decode %.4f" % ((time.perf_counter() - t0) / 20 * 1e3))
