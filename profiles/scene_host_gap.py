"""Where the host time of a one-at-a-time scene goes: Python before the C call, the C call, Python after it, against the
GPU time between the scene's first and last event.  PYTHONPATH=. python profiles/scene_host_gap.py"""
import time, numpy as np, torch
import bench
from canonicalvoting_amd import _lib, pipeline
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.hough import HoughVoting

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = _lib.lib()
torch.manual_seed(0)
model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().eval()
hv = HoughVoting(bench.RES, bench.NUM_ROTS)
scenes = [bench.ResidentScene(seed, 80000, dev) for seed in range(4)]
real = L.cv_detect_scene_f32
marks = []


def wrapped(*a):
    t0 = time.perf_counter()
    rc = real(*a)
    marks.append((t0, time.perf_counter()))
    return rc


L.cv_detect_scene_f32 = wrapped
st = torch.cuda.Stream(dev)
rows = []
with torch.cuda.stream(st):
    for k in range(60):
        ev = bench.step_events()
        s = scenes[k % 4]
        t_in = time.perf_counter()
        bench.run_step(model, hv, s, ev, True)
        t_out = time.perf_counter()
        c0, c1 = marks[-1]
        st.synchronize()
        rows.append((1e6 * (c0 - t_in), 1e6 * (c1 - c0), 1e6 * (t_out - c1), 1e3 * ev[0].elapsed_time(ev[4])))
r = np.array(rows[20:])
print("python before the C call %.0f us | C call %.0f us | python after %.0f us | GPU first->last event %.0f us | C call - GPU %.0f us"
      % (r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), (r[:, 1] - r[:, 3]).mean()))
