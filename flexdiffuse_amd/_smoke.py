'''One tiny end-to-end invocation of the hot path on a HIP device, checked against the CPU
oracle (called from __graft_entry__.smoke(); the oracle is only the checker).'''
import torch


def run(dev) -> None:
    from flexdiffuse_amd import Guide, SimpleGuide, build
    from flexdiffuse_amd.encode.clip import CLIPEncoder
    from oracle import clip_ref, guide_ref, pipeline_ref
    sds = build.synthetic_state_dicts('mini', seed=0)
    sds = {k: {n: t.half().float() for n, t in sd.items()} for k, sd in sds.items()}
    pipe, clip, tok = build.build_models(sds, 'mini', dev)
    ucfg, vcfg, ccfg = build.configs('mini')
    enc = CLIPEncoder(clip, tok)
    prompt = 'a photo of a turtle'
    steps, guidance, hw = 4, 8.0, 64
    embeds = Guide(clip, tok, device='cuda').embeds(prompt=prompt)
    out = pipe(guide=SimpleGuide(enc, pipe.unet, guidance, steps, embeds), init_size=(hw, hw),
               generator=torch.Generator('cpu').manual_seed(7), output_type='np')
    emb_ref = clip_ref.text_hidden(sds['clip'], ccfg, tok(prompt).input_ids)
    unc_ref = clip_ref.text_hidden(sds['clip'], ccfg, tok('').input_ids)
    lat0 = torch.randn((1, 4, hw // 8, hw // 8), generator=torch.Generator('cpu').manual_seed(7))
    lat_ref, used = pipeline_ref.denoise(sds['unet'], ucfg, emb_ref, unc_ref, lat0, steps, guidance)
    img_ref = pipeline_ref.decode_image(sds['vae'], vcfg, lat_ref)
    assert used == [int(t) for t in pipe.scheduler.timesteps], 'timestep lists differ'
    p = pipeline_ref.psnr(pipe.last_images.cpu(), img_ref)
    assert p >= 40.0, f'final-image PSNR {p:.1f} dB < 40 dB vs CPU oracle'
    print(f'smoke ok: mini pipeline ({steps} DDIM steps, CFG {guidance}) PSNR {p:.1f} dB vs CPU oracle, '
          f'{len(out.images)} image(s) {out.images[0].shape}')
