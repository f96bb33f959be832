import torch, sys, os
sys.path.insert(0, os.getcwd())
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
x = torch.randn((8, 4, 64, 64), device=dev)
w = torch.randn((320, 4, 3, 3)) * 0.1
nw = ops.prep_conv_narrow(w, torch.zeros(320), dev)
cw = ops.prep_conv(w, torch.zeros(320), dev, cin_pad=8)
buf = torch.empty((16 * 4096, 960), dtype=torch.float16, device=dev)
view = buf[:, 640:]
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('direct, 2 replicas into a concat slice: %.1f us' % t(lambda: ops.conv3x3_narrow(x, nw, out2=view, rep2=2)))
print('direct, no replicas: %.1f us' % t(lambda: ops.conv3x3_narrow(x, nw)))
def old():
    h = ops.conv2d(ops.nchw_to_nhwc(x, c_pad=8), cw)
    ops.repeat_rows(h.t, 2, out=view)
print('layout + im2col + GEMM + fan-out copy: %.1f us' % t(old))
