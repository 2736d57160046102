"""vote op alone under scene concurrency: T threads x N runs on their own streams, every grid compared bit for bit with
the one-at-a-time result (which the GPU tests pin to the oracle).  Prints how many runs differ and where."""
import os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions
dev = torch.device("cuda:0")
T, N = int(os.environ.get("T", "8")), int(os.environ.get("N", "200"))
LOAD = os.environ.get("LOAD", "0") == "1"
scenes = []
for seed in range(4):
    sc = make_scene(seed, n_points=1500 + 250 * seed, res=0.06, room=(2.0, 1.0, 2.0), n_boxes=3, margin=0.6, box_scale=0.5)
    xyz, scale, prob, cls = synth_predictions(sc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    scenes.append((t(sc.points), t(xyz), t(scale), t(prob)))
hv0 = HoughVoting(0.06, 120)
with torch.no_grad():
    ref = [[g.clone() for g in hv0(*s)] for s in scenes]
torch.cuda.synchronize()
bad = []
def worker(i):
    hv = HoughVoting(0.06, 120)
    junk = torch.randn(2048, 2048, device=dev)
    with torch.cuda.stream(torch.cuda.Stream(dev)), torch.no_grad():
        for k in range(N):
            s = scenes[(k + i) % 4]
            if LOAD:
                junk = (junk @ junk).clamp_(-1, 1)          # other kernels on this stream between the votes
            g = hv(*s)
            r = ref[(k + i) % 4]
            if not all(torch.equal(a, b) for a, b in zip(g, r)):
                d = (g[0] != r[0]).nonzero()
                bad.append((i, k, (k + i) % 4, int(d.shape[0]), d[:4].tolist(), float((g[0] - r[0]).abs().max()),
                            float(g[0].double().sum() - r[0].double().sum())))
threads = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
[t.start() for t in threads]; [t.join() for t in threads]
print("T=%d N=%d LOAD=%s: %d of %d runs differ" % (T, N, LOAD, len(bad), T * N))
for b in bad[:8]:
    print("  thread %d run %d scene %d: %d cells differ %s max|d| %.4g sum diff %.6g" % b)
