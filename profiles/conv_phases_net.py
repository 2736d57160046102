"""conv_rows phase ticks (CV_CONV_PROF=1 twin) summed over every launch of one fused MinkUNet34C forward."""
import os, sys, re, subprocess
if len(sys.argv) > 1 and sys.argv[1] == 'run':
    os.environ['CV_CONV_PROF'] = '1'
    os.environ['CV_NET_PROGRAM'] = '0'
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
    with torch.no_grad():
        m(ME.SparseTensor(f, c4, device=dev))
        print('=== second forward', file=sys.stderr, flush=True)
        m(ME.SparseTensor(f, c4, device=dev))
    torch.cuda.synchronize()
else:
    out = subprocess.run([sys.executable, __file__, 'run'], capture_output=True, text=True).stderr
    out = out.split('=== second forward')[1]
    tot = {}
    waves_total = 0
    by_level = {}
    for line in out.splitlines():
        m = re.match(r'conv_rows<(\d)> n_out (\d+) cin (\d+) cout (\d+) K (\d+) splits (\d+): waves (\d+), ticks/wave:(.*)', line)
        if not m:
            continue
        waves = int(m.group(7))
        toks = m.group(8).split()
        vals = {}
        i = 0
        while i < len(toks):
            vals[toks[i]] = float(toks[i + 1]); i += 2
        lv = by_level.setdefault(int(m.group(2)), {})
        for k, v in vals.items():
            tot[k] = tot.get(k, 0.0) + v * waves
            lv[k] = lv.get(k, 0.0) + v * waves
        waves_total += waves
    s = sum(tot.values())
    print('share of wave time over all conv_rows launches of one forward (%d waves):' % waves_total)
    for k, v in tot.items():
        print('  %-18s %5.1f %%' % (k, 100 * v / s))
    for n, lv in sorted(by_level.items(), reverse=True):
        sl = sum(lv.values())
        print('rows %6d: %4.1f %% of all wave time |' % (n, 100 * sl / s), ' '.join('%s %.0f%%' % (k[:9], 100 * v / sl) for k, v in lv.items()))
