'''A/B of the V^T projection tile (FD_GEMM_VT_TILE = 0: 128x64 / 4 waves, 9: 128x160 / 8 waves at
M >= 8192) on the UNet's self-attention V shapes, with and without the LayerNorm fold; each arm is a
child process (the knob is read once).  The 256x160 / 16-wave arm of the recorded run
(profiles/r02_session_ab.txt) was removed from the library afterwards.
    python tools/ab_vt.py'''
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    from flexdiffuse_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for (B, HW, C) in ((16, 4096, 320), (16, 1024, 640), (16, 256, 1280), (16, 64, 1280)):
        M = B * HW
        x = torch.randn((M, C), device=dev).half()
        w = torch.randn((C, C)) / C ** 0.5
        b = torch.randn(C) * 0.1
        gamma, beta = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
        ldv = (HW + 7) // 8 * 8
        for fold in (0, 1):
            if fold:
                lw = ops.prep_linear_ln(w, b, gamma, beta, dev)
                st = ops.ln_row_stats(x)
                run = lambda: ops.gemm_vt(x, lw, B, HW, ldv, ln_stats=st)
                xn = torch.nn.functional.layer_norm(x.float(), (C,), gamma.to(dev), beta.to(dev))
                ref = xn @ w.to(dev).t() + b.to(dev)
            else:
                lw = ops.prep_linear(w, b, dev)
                run = lambda: ops.gemm_vt(x, lw, B, HW, ldv)
                ref = x.float() @ w.to(dev).half().float().t() + b.to(dev)
            out = run()
            got = out.float().view(B, C, ldv)[:, :, :HW].transpose(1, 2).reshape(M, C)
            err = (got - ref).abs().max().item()
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            print(f'  M={M:6d} C={C:5d} fold={fold}: {us:7.1f} us  {2.0 * M * C * C / us / 1e6:6.0f} TFLOP/s  max|err| {err:.4f}', flush=True)
    sys.exit(0)
for arm in ('0', '9'):
    env = dict(os.environ, FD_GEMM_VT_TILE=arm)
    print(f'FD_GEMM_VT_TILE={arm}', flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, check=False)
