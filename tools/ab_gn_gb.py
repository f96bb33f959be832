'''GroupNorm slab kernel: groups per workgroup (GB) A/B via the temporary FD_GN_GB_MIN knob (child processes).'''
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import torch
    from flexdiffuse_amd import ops
    dev = torch.device('cuda:0')
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = []
    for (B, HW, C) in [(16, 1024, 640), (16, 1024, 1280), (16, 1024, 1920), (16, 1024, 960), (16, 256, 1280), (16, 256, 2560), (16, 256, 1920), (16, 64, 1280), (16, 64, 2560)]:
        x = ops.Act(torch.randn((B * HW, C), device=dev).half(), B, HW, 1)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        ms = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True))
        out.append(f'{ms * 1e3:5.1f}us/{B * HW * C * 4 / ms / 1e9:4.2f}TB/s')
    print(f"GB_MIN {os.environ.get('FD_GN_GB_MIN', '1')} SLAB_OFF {os.environ.get('FD_GN_SLAB_OFF', '0')} SWAP {os.environ.get('FD_GN_SWAP', '0')}: " + ' '.join(out), flush=True)
    sys.exit(0)
for gb, off, sw in (('1', '0', '0'), ('1', '0', '1'), ('4', '0', '0'), ('4', '0', '1'), ('8', '0', '0'), ('8', '0', '1'), ('1', '1', '0'), ('1', '2', '0')) * 2:
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=dict(os.environ, FD_GN_GB_MIN=gb, FD_GN_SLAB_OFF=off, FD_GN_SWAP=sw))
