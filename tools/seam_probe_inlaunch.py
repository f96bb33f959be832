'''Round 6, VERDICT r5 next 1(b): the cooperative in-launch split-K reduction (fd_gemm_desc.sk_sync, EXPERIMENTAL, ops.SPLITK_INLAUNCH) against the
product chain, same process, interleaved, same method as tools/seam_probe_gn.py.  Chains, at the 8x8 and 16x16 levels of the bench forward:
  product      split-K partial pass -> finish pass with the GroupNorm + SiLU inside (2 launches)
  in-launch    partial pass that finishes its own tiles (arrival counters, agent-scope release / acquire) -> GroupNorm launch (2 launches)
  round 5      partial pass -> finish pass -> GroupNorm launch (3 launches)
and the same without a GroupNorm behind the convolution (conv + finish vs conv alone).  Outputs are compared bit for bit.
    python tools/seam_probe_inlaunch.py'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def chain_time(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    return best


for (B, H, C) in ((16, 8, 1280), (16, 16, 1280)):
    M = B * H * H
    x = ops.Act((torch.randn((M, C), generator=g) * 0.7).half().to(dev), B, H, H)
    cw = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5, torch.randn(C, generator=g), dev)
    spec = ops.GNSpec(torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5, True)
    kw = dict(bias2=torch.randn((B, C), generator=g).to(dev), ld_bias2=C)

    def arm(inl, fuse, gn):
        ops.SPLITK_INLAUNCH, ops.GN_FINISH_FUSE = inl, fuse
        return (lambda: ops.conv2d(x, cw, gn=spec, keep=False, **kw)) if gn else (lambda: ops.conv2d(x, cw, **kw))
    arms = {'product (finish + GN in one pass)': (False, True, True), 'in-launch reduction + GN launch': (True, True, True),
            'round 5 (finish, GN: three launches)': (False, False, True), 'conv + finish, no GN': (False, True, False), 'conv in-launch, no GN': (True, True, False)}
    # correctness first: the in-launch forms give the bits of the finish launch
    ref_gn = arm(False, False, True)()[1].t.clone()
    got_gn = arm(True, True, True)()[1].t.clone()
    ref_c = arm(False, True, False)().t.clone()
    got_c = arm(True, True, False)().t.clone()
    torch.cuda.synchronize()
    same = [torch.equal(ref_gn, got_gn), torch.equal(ref_c, got_c)]
    bad = 0
    fn = arm(True, True, False)
    for _ in range(200):
        bad += int((fn().t != ref_c).sum())
    res = {k: [] for k in arms}
    for rnd in range(4):
        for k, a in arms.items():
            res[k].append(chain_time(arm(*a)))
    ops.SPLITK_INLAUNCH, ops.GN_FINISH_FUSE = False, True
    print(f'{H}x{H} (M {M}, N {C}, K {9 * C}): in-launch output equals the finish launch: GN chain {same[0]}, conv {same[1]}; 200 repeats, mismatching elements: {bad}')
    for k, v in res.items():
        s = sorted(v)
        print(f'   {k:40s} {" ".join(f"{t:.1f}" for t in v)} us -> median {0.5 * (s[1] + s[2]):.1f}', flush=True)
