"""scenes-in-flight finding of the vote tile kernel: the vote on captured network predictions, repeated on one stream while six
other streams run a synthetic co-resident load (profiles/microbench/lds_hammer.hip) instead of the convolutions - which
ingredient disturbs it?  Run with the tile kernel built in the shape that goes wrong next to the convolutions
(CV_HV_DEFS="-DHV_TX=16 -DHV_TW=8")."""
import ctypes, os, sys, threading
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from canonicalvoting_amd import pipeline
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene
H = ctypes.CDLL(os.path.join(HERE, "microbench", "liblds_hammer.so"))
H.lds_hammer_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = MinkUNet34C(3, 64).to(dev).eval()
ins = []
for seed in range(4):
    sc = make_scene(seed, n_points=1500 + 250 * seed, res=0.06, room=(2.0, 1.0, 2.0), n_boxes=3, margin=0.6, box_scale=0.5)
    c4 = torch.cat([torch.zeros((len(sc.coords), 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
    f = (torch.from_numpy(sc.feats) * 2 - 1).to(dev)
    with torch.no_grad():
        y = model(ME.SparseTensor(f, c4, device=dev))
        xyz, scale, prob, cls = pipeline.head_joint(y.F)
        ins.append(((c4[:, 1:] * 0.06).float().contiguous(), xyz.clone(), scale.clone(), prob.clone()))
hv0 = HoughVoting(0.06, 120)
with torch.no_grad():
    ref = [[g.clone() for g in hv0(*s)] for s in ins]
torch.cuda.synchronize()
sink = torch.zeros(65536, device=dev)
src = torch.rand(1 << 24, device=dev)
stop = False
def load(mode):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(8):
                H.lds_hammer_launch(mode, 768, 200, sink.data_ptr(), src.data_ptr(), src.numel(), ctypes.c_void_p(st.cuda_stream))
            st.synchronize()
NAMES = {0: "nothing", 1: "LDS b128 traffic", 2: "LDS b32 traffic", 3: "fp16 MFMA only", 4: "fp32 MFMA only", 5: "LDS b128 + fp16 MFMA",
         6: "LDS u64 atomics", 7: "global loads"}
for mode in [int(m) for m in (sys.argv[1:] or ["0", "1", "2", "3", "4", "5", "6", "7"])]:
    stop = False
    bg = [threading.Thread(target=load, args=(mode,)) for _ in range(4 if mode else 0)]
    [t.start() for t in bg]
    bad = 0
    hv = HoughVoting(0.06, 120)
    with torch.cuda.stream(torch.cuda.Stream(dev)), torch.no_grad():
        for k in range(200):
            g = hv(*ins[k % 4])
            bad += 0 if torch.equal(g[0], ref[k % 4][0]) else 1
    stop = True
    [t.join() for t in bg]
    torch.cuda.synchronize()
    print("co-resident load %-22s: %3d of 200 votes differ from the one-at-a-time result" % (NAMES[mode], bad), flush=True)
