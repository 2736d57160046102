"""Wall-time ablations of conv_rows_x6 (default) or, with ABLATE_FP32=1, of the fp32-MFMA conv_rows (instrumented
twin with the counters off); CV_CONV_DBG bits: 1 no MFMA, 2 no gathers, 4 no weight loads, 8 no epilogue, 16 no operand split, 32 no staging,
64 / 128 B / A operand fragments read from LDS once per unit instead of per MFMA group."""
import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from canonicalvoting_amd import me as ME
    from canonicalvoting_amd.synth import make_scene
    dev = torch.device('cuda')
    PIECES = int(os.environ.get('ABLATE_PIECES', '2'))      # 2: fp16 pairs (the eval program's kernel), 3: bf16 triples
    sc = make_scene(3, 80000)
    c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
    cm = ME.CoordinateManager(c4).fused_plan()[0]
    out = []
    for ts, cin, cout in ((1, 96, 96), (2, 96, 96), (4, 128, 128), (8, 256, 256), (16, 256, 256)):
        n = cm.num_rows(ts)
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.02
        nbr = cm.kernel_map(3, ts)
        if n >= 16384:
            perms = cm.mask_perms(3, ts, 4)
            fn = lambda: ME.conv_forward_masked(x, w, nbr, perms, n, relu=True, pieces=PIECES)
        else:
            fn = lambda: ME.conv_forward(x, w, nbr, n, relu=True, pieces=PIECES)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append('%8.1f' % (e0.elapsed_time(e1) / 20 * 1e3))
    print('dbg %2s: ' % os.environ.get('CV_CONV_DBG', '0') + ' '.join(out))
else:
    print('us per conv (+finish): ts1 96>96  ts2 96>96  ts4 128>128  ts8 256>256  ts16 256>256')
    for dbg in (0, 1, 64, 192, 193, 32, 15, 47):     # 64 / 128: B / A operand fragments read from LDS once per unit
        env = dict(os.environ, CV_CONV_DBG=str(dbg), CV_NET_PROGRAM='0')
        if os.environ.get('ABLATE_FP32'):
            env.update(CV_CONV_PROF='q', CV_CONV_X6='0')      # the fp32-MFMA kernel (instrumented twin, counters off)
        subprocess.run([sys.executable, __file__, 'run'], env=env)
