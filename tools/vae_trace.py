'''Five VAE decodes at the headline shape (8 images, 64x64 latents -> 512x512) for
`rocprofv3 --kernel-trace --stats -- python3 tools/vae_trace.py` (per-kernel time of the decode).'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build
from flexdiffuse_amd.vae import AutoencoderKL
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('vae',))
vae = AutoencoderKL(sds['vae'], build.configs('sd15')[1], dev, encoder=False)
z = torch.randn((8, 4, 64, 64), device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(2):
    vae.decode(z).sample
torch.cuda.synchronize(); t0 = time.time()
for i in range(n):
    vae.decode(z).sample
torch.cuda.synchronize()
print(f'VAE decode of 8 images: {1e3 * (time.time() - t0) / n:.2f} ms', flush=True)
