'''Deep-level convolutions (split-K): flat 1-D grid with slice = id % split_k (a K slice's workgroups share an XCD,
FD_GEMM_SK_FLAT=1, default) against the 2-D grid (FD_GEMM_SK_FLAT=0).  Arms are child processes (the knob is read once);
prints us per conv (partial pass + finish kernel) for the 8x8 / 16x16-level shapes of the SD1.5 UNet at CFG batch 16,
and checks that both arms give bit-identical outputs (same partial sums, same fixed-order reduction).'''
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import torch
    from flexdiffuse_amd import ops
    dev = torch.device('cuda:0')

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n
    out = []
    for (B, H, Cin, Cout) in [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280),
                             (16, 16, 1920, 1280), (16, 16, 640, 1280)]:
        g = torch.Generator().manual_seed(Cin + H)
        x = ops.Act((torch.randn((B * H * H, Cin), generator=g) * 0.7).half().to(dev), B, H, H)
        w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5, torch.randn(Cout, generator=g), dev)
        us = timeit(lambda: ops.conv2d(x, w))
        y = ops.conv2d(x, w).t
        h = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:8]
        out.append(f'{H}x{H} {Cin}->{Cout}: {us:.1f} us ({2 * B * H * H * Cout * 9 * Cin / us / 1e6:.0f} TF) #{h}')
    print('   ' + ' | '.join(out), flush=True)
    sys.exit(0)
# arms: "VAR=val[,VAR2=val2]" per argument (default: the flat-grid A/B), e.g.  python tools/ab_splitk.py FD_GEMM_PF=0 FD_GEMM_PF=2
for arm in tuple(sys.argv[1:] or ('FD_GEMM_SK_FLAT=1', 'FD_GEMM_SK_FLAT=0')) * 2:
    print(arm, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child'],
                   env=dict(os.environ, **dict(kv.split('=', 1) for kv in arm.split(','))), check=False)
