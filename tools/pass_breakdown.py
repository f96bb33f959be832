'''Where one bench pass spends its time outside the UNet loop: Guide.embeds, 50-step loop,
VAE decode, device->host copy of the images (synchronised sections; 3 repetitions).'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from flexdiffuse_amd import Guide, SimpleGuide, build, dist as fdist, ops
from flexdiffuse_amd.encode.clip import CLIPEncoder
dev = torch.device('cuda:0')
PRESET = sys.argv[1] if len(sys.argv) > 1 else 'sd15'   # python tools/pass_breakdown.py [preset] [image size]
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sds = build.synthetic_state_dicts(PRESET, seed=0)
pipe, clip, tok = build.build_models(sds, PRESET, dev, vae_encoder=False)
g = Guide(clip, tok, device='cuda'); enc = CLIPEncoder(clip, tok)
prompts = bench.synth_prompts(8); img = bench.synth_image(2, 512, 512)
noise = fdist.global_noise(8, (4, SIZE // 8, SIZE // 8), 1337).to(dev)
def sync(): torch.cuda.synchronize(); return time.time()
for rep in range(3):
    t0 = sync()
    emb = g.embeds(prompt=prompts, guide=img, **bench.GUIDANCE['linear'])
    t1 = sync()
    sg = SimpleGuide(enc, pipe.unet, 8.0, 50, emb)
    t2 = sync()
    # the loop alone: call the pipeline pieces by hand
    pipe.scheduler.set_timesteps(50)
    lat = noise.clone()
    for t in pipe.scheduler.timesteps:
        eps = pipe.unet.forward_nhwc(lat, int(t), sg.stacked_embeds(), rep=2)
        ops.cfg_ddim_step(lat, eps, 8, 4, (SIZE // 8) ** 2, True, 8.0, pipe.scheduler.step_coefficients(int(t))[:4], False)
    t3 = sync()
    im = pipe.decode_latents(lat)
    t4 = sync()
    host = im.cpu().permute(0, 2, 3, 1).numpy()
    t5 = sync()
    print(f'embeds {1e3*(t1-t0):.1f} ms | SimpleGuide ctor (uncond prompt) {1e3*(t2-t1):.1f} | 50-step loop {1e3*(t3-t2):.1f} '
          f'| VAE decode {1e3*(t4-t3):.1f} | D2H + numpy {1e3*(t5-t4):.1f} | total {1e3*(t5-t0):.1f}', flush=True)
