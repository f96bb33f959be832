'''CPU oracle: the control flow of the reference's `FlexPipeline` around duck-typed model objects.

TEST INFRASTRUCTURE ONLY.  A restatement of pipeline/flex.py:46-83 (constructor: set_format,
the `steps_offset` 0 -> 1 config rewrite with its DeprecationWarning), :112-124 (decode scaling,
clamp, NHWC numpy / PIL) and :126-310 (`__call__`: strength check, timesteps, img2img encode ->
posterior sample -> scale -> repeat -> init_timestep / add_noise level -> t_start, txt2img randn,
LMS sigma scaling, eta forwarding by signature, loop with LMS t_index, debug latent list, tuple /
record return) that calls whatever vae / scheduler / guide objects it is given.

PINNED by tests/golden/flexcall_goldens.npz: the reference's own file run with the recording
stubs of tests/flexcall_stubs.py (tests/golden/make_flexcall_goldens.py); the CPU test
tests/test_oracle_flexcall.py replays the same stubs through this file and requires the same
call trace and images.  The specialised loops of oracle/pipeline_ref.py (`denoise`,
`img2img_init`) are checked against the same traces there.
'''
import inspect
import warnings

import numpy as np
import torch

VAE_SCALE = 0.18215


class DDIMSchedulerRef():
    pass


class PNDMSchedulerRef():
    pass


class LMSDiscreteSchedulerRef():
    pass


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


class Output():
    def __init__(self, images, nsfw_content_detected):
        self.images, self.nsfw_content_detected = images, nsfw_content_detected


def numpy_to_pil(images):
    from PIL import Image
    if images.ndim == 3:
        images = images[None, ...]
    return [Image.fromarray(im) for im in (images * 255).round().astype('uint8')]


class FlexPipelineRef():
    def __init__(self, vae, clip, tokenizer, unet, scheduler, preprocess=None):
        scheduler = scheduler.set_format('pt')                       # :55
        cfg = scheduler.config
        if hasattr(cfg, 'steps_offset') and cfg['steps_offset'] != 1:  # :57-70
            warnings.warn('scheduler config is outdated: steps_offset should be 1', DeprecationWarning)
            fixed = dict(cfg)
            fixed['steps_offset'] = 1
            scheduler._internal_dict = _Cfg(fixed)
        self.vae, self.clip, self.tokenizer, self.unet, self.scheduler = vae, clip, tokenizer, unet, scheduler
        if preprocess is None:
            from .clip_ref import preprocess
        self.preprocess = preprocess
        self.device = torch.device('cpu')

    def _latents_to_image(self, latents, pil=True):                  # :112-124
        image = self.vae.decode(1 / VAE_SCALE * latents).sample
        image = (image / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).numpy()
        return numpy_to_pil(image) if pil else image

    @torch.no_grad()
    def __call__(self, guide, init_image=None, init_size=(512, 512), strength=0.6, eta=0.0,
                 generator=None, output_type='pil', return_dict=True, debug=False):
        if not 0 <= strength <= 1:                                   # :170-172
            raise ValueError(f'The value of strength should in [0.0, 1.0] but is {strength}')
        B, steps, sch = guide.batch_size, guide.steps, self.scheduler
        lms = isinstance(sch, LMSDiscreteSchedulerRef)
        sch.set_timesteps(steps)                                     # :177
        if init_image is not None:                                   # :181 (E8: `is not None`)
            if not isinstance(init_image, torch.Tensor):
                init_image = self.preprocess(init_image)
            z = self.vae.encode(init_image).latent_dist.sample(generator=generator)   # :189-191
            z = torch.cat([VAE_SCALE * z] * B)                       # :192-194
            offset = sch.config.get('steps_offset', 0)               # :197
            init_timestep = min(int(steps * strength) + offset, steps)
            level = steps - init_timestep if lms else sch.timesteps[-init_timestep]   # :200-209
            level = torch.tensor([level] * B, dtype=torch.long)
            noise = torch.randn(z.shape, generator=generator)        # :212-214
            x = sch.add_noise(z, noise, level)                       # :215
            t_start = max(steps - init_timestep + offset, 0)         # :221
        else:
            h, w = init_size
            x = torch.randn((B, self.unet.in_channels, h // 8, w // 8), generator=generator)   # :226-230
            sch.set_timesteps(steps)                                 # :233
            if lms:
                x = x * sch.sigmas[0]                                # :236-238
            t_start = 0
        extra = {'eta': eta} if 'eta' in inspect.signature(sch.step).parameters else {}   # :247-251
        history = [x] if debug else None                             # :254-256
        for i, t in enumerate(sch.timesteps[t_start:]):              # :262-287
            t_index, model_in = t, x
            if lms:
                t_index = t_start + i
                sigma = sch.sigmas[t_index]
                model_in = x / ((sigma ** 2 + 1) ** 0.5)
            eps = guide.noise_pred(model_in, t)
            x = sch.step(eps, t_index, x, **extra).prev_sample
            if history:
                history.append(x)
        self.last_latents = x
        if history:                                                  # :289-301
            batches = [self._latents_to_image(l, output_type == 'pil') for l in history]
            images = [im for b in batches for im in b] if isinstance(batches[0], list) \
                else np.concatenate(batches, axis=0)
        else:
            images = self._latents_to_image(x, output_type == 'pil')
        if not return_dict:                                          # :305-306
            return (images, False)
        return Output(images, [False for _ in images])
