"""vote op on the teacher predictions of bench scene 0 (or --net: the random-init network's), HIP-event time of the
whole op; run under rocprofv3 for the per-kernel split (profiles/vote_prof.sh); --ticks prints the phase profile"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import hv_cuda
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions
dev = torch.device("cuda:0")
large = "--large" in sys.argv
sc = make_scene(0, n_points=300000, room=(9.0, 3.0, 9.0), n_boxes=40) if large else make_scene(0, n_points=80000)
xyz, scale, prob, cls = synth_predictions(sc)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p, x, s, o = t(sc.points), t(xyz), t(scale), t(prob)
hv = HoughVoting(0.03, 120)
if "--ticks" in sys.argv:
    with torch.no_grad():
        for _ in range(3):
            hv(p, x, s, o)
        torch.cuda.synchronize()
        hv_cuda.set_algorithm(24)
        for _ in range(2):
            hv(p, x, s, o)
        torch.cuda.synchronize()
    hv_cuda.set_algorithm(0)
with torch.no_grad():
    for _ in range(3):
        g = hv(p, x, s, o)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for i in range(20):
        ev[i].record()
        g = hv(p, x, s, o)
    ev[20].record()
torch.cuda.synchronize()
print("vote", "300k" if large else "80k", "event ms/op %.4f" % (ev[0].elapsed_time(ev[20]) / 20))
