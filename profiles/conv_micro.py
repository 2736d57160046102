"""One ts1-sized masked 3x3x3 conv (96->96, 80k rows, 4 mask groups) repeated - for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
sc = make_scene(3, 80000)
c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]
x = torch.randn(80000, 96, device=dev)
w = torch.randn(27, 96, 96, device=dev) * 0.02
nbr = cm.kernel_map(3, 1)
perms = cm.mask_perms(3, 1, 4)
HL = os.environ.get('MICRO_HL', '0') == '1'          # hl-format operands (conv_hl)
if HL:
    x = ME.to_hl(x)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    y = ME.conv_forward_masked(x, w, nbr, perms, 80000, relu=True, pieces=int(os.environ.get('MICRO_PIECES', '2')),
                               in_hl=HL, out_hl=HL)
torch.cuda.synchronize()
print(float(y.abs().mean()))
