import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
sc = make_scene(3, 80000)
c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
f = (torch.from_numpy(sc.feats) * 2 - 1).to(dev)
torch.manual_seed(0)
m = MinkUNet34C(3, 64).cuda().eval()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e3
def plan_only():
    x = ME.SparseTensor(f, c4, device=dev)
    cm, sm, om = x.coordinate_manager.fused_plan()
    for ts in (1, 2, 4, 8, 16): cm.kernel_map(3, ts)
    for ts in (1, 2, 4, 8): cm.kernel_map(2, ts, 2); cm.up_map(2 * ts); cm.up_perm(2 * ts)
    for ts in (1, 2): cm.mask_perms(3, ts, m.MASK_GROUPS)
    return x
with torch.no_grad():
    print('plan only ms %.3f' % t(plan_only))
    x = plan_only()
    print('net with cached plan ms %.3f' % t(lambda: m(x)))
    print('net incl. plan ms %.3f' % t(lambda: m(ME.SparseTensor(f, c4, device=dev))))
