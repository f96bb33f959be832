'''UNet-forward A/B over ENVIRONMENT knobs (runtime or library), measured on the launch-plan replay of the full-size SD1.5 forward (CFG batch 16):
each argument is one arm, "VAR=val[,VAR2=val2]" ("-" = defaults); arms run as child processes, the list is repeated twice.
    python tools/ab_env_plan.py - HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0'''
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import torch
    from flexdiffuse_amd import build, hip
    from flexdiffuse_amd.unet import UNet2DConditionModel
    dev = torch.device('cuda:0')
    sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
    unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
    x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
    t_dev = torch.full((1,), 400.0, device=dev)
    unet.forward_nhwc(x, t_dev, ctx, rep=2)
    pool = torch.cuda.MemPool(); plan = hip.Plan()
    with torch.cuda.use_mem_pool(pool, device=dev), plan.record():
        unet.forward_nhwc(x, t_dev, ctx, rep=2)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan.replay()
    out = []
    for mode, fn in (('plan', plan.replay), ('graph', graph.replay)):
        res = []
        for rep in range(3):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(20): fn()
            torch.cuda.synchronize()
            res.append(f'{1e3 * (time.time() - t0) / 20:.3f}')
        out.append(f'{mode} ' + ' '.join(res))
    print(f'   UNet forward (CFG batch 16, {len(plan)} launches) ms: ' + ' | '.join(out), flush=True)
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
