'''UNet-forward A/B over environment knobs: each argument is one arm, "VAR=val[,VAR2=val2]" ("-" = defaults);
arms run as child processes (knobs are read once per process), the list is repeated twice.
    python tools/ab_env.py - FD_CONV_TAPFAST=2 FD_GEMM_PERSIST=0'''
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import torch
    from flexdiffuse_amd import build
    from flexdiffuse_amd.unet import UNet2DConditionModel
    dev = torch.device('cuda:0')
    sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
    unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
    x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
    out = []
    for rep in range(3):
        for _ in range(3): unet.forward_nhwc(x, 400, ctx, rep=2)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): unet.forward_nhwc(x, 400, ctx, rep=2)
        torch.cuda.synchronize()
        out.append(f'{1e3 * (time.time() - t0) / 20:.3f}')
    print('   UNet forward (CFG batch 16) ms:', ' '.join(out), flush=True)
    sys.exit(0)
arms = sys.argv[1:] or ['-']
for arm in arms * 2:
    env = dict(os.environ)
    if arm != '-':
        for kv in arm.split(','):
            k, v = kv.split('=', 1)
            env[k] = v
    print(arm, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, check=False)
