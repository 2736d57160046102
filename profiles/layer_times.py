"""Per-conv-call times of the fused MinkUNet34C forward on one 80k-point scene (HIP events around every
cv_sp_conv_f32 call, averaged over repeats): which layers the net forward's milliseconds go to."""
import os, sys
os.environ['CV_NET_PROGRAM'] = '0'      # per-conv timing needs the Python-issued forward
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene

dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
sc = make_scene(3, N)
c4 = torch.cat([torch.zeros((N, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
f = (torch.from_numpy(sc.feats) * 2 - 1).to(dev)
torch.manual_seed(0)
m = MinkUNet34C(3, 64).cuda().eval()
x = ME.SparseTensor(f, c4, device=dev)
orig = ME.conv_forward
records = []
REPS = 10


PIECES = int(os.environ.get("CV_LAYER_PIECES", "2"))      # 2: fp16 pairs (the default program's format), 3: bf16 triples


def timed(x_feats, weight, nbr, n_out, **kw):
    kw.setdefault("pieces", PIECES)
    w = weight if weight.dim() == 3 else weight[None]
    K, cin, cout = w.shape
    for _ in range(2):
        out = orig(x_feats, weight, nbr, n_out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        out = orig(x_feats, weight, nbr, n_out, **kw)
    e1.record()
    torch.cuda.synchronize()
    pairs = int((nbr >= 0).sum()) if nbr is not None else n_out
    records.append((n_out, K, cin, cout, kw.get('perm_groups', 0), e0.elapsed_time(e1) / REPS * 1e3, pairs))
    return out


with torch.no_grad():
    m(x)
    ME.conv_forward = timed
    m(x)
    ME.conv_forward = orig
tot = 0.0
print('%3s %7s %4s %4s %4s %3s %9s %9s %8s %7s' % ('#', 'n_out', 'K', 'cin', 'cout', 'grp', 'us', 'pairs', 'TF/s', 'live%'))
for i, (n_out, K, cin, cout, g, us, pairs) in enumerate(records):
    tot += us
    print('%3d %7d %4d %4d %4d %3d %9.1f %9d %8.2f %7.1f' % (i, n_out, K, cin, cout, g, us, pairs,
                                                          2.0 * pairs * cin * cout / us * 1e-6, 100.0 * pairs / (n_out * K)))
print('total conv us %.1f' % tot)
