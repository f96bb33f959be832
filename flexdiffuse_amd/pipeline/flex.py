'''Image-guided denoising pipeline -- host-side mirror of the reference's
`pipeline/flex.py` `FlexPipeline` (ctor :46-83, attention slicing :85-110,
_latents_to_image :112-124, __call__ :126-310) with the same call signature, return types
and ValueError, driving the gfx950 kernels.

Device loop: for `SimpleGuide` + DDIM the whole step is  UNet(NHWC fp16, CFG batch built
inside the layout kernel) -> fused CFG + DDIM update on the fp32 latents  with no host
round trip; any other guide object goes through the reference protocol
(`guide.noise_pred` + `scheduler.step`) unchanged.  The UNet forward of the fused loop is
replayed from a captured HIP GRAPH (`use_graph`, default: one host call per denoising step;
the mode `bench.py` measures) and, if capture fails on a box, from a LAUNCH PLAN (`use_plan`):
its ~310 C-ABI launches recorded once per (shape, context) and re-issued each step by one
library call -- the same eager launches in the same order, without the Python front's per-op
work.  Same kernels, same order, bit-identical results in all three modes.

Deliberate differences: E6 initial noise is drawn on the generator's own
device -- pass a CPU generator for results independent of the GPU count; E8 `init_image`
is tested with `is not None`.
'''
from __future__ import annotations

import contextlib
import gc
import inspect
import warnings
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .. import hip, ops
from ..encode.clip import preprocess
from ..scheduler import DDIMScheduler, LMSDiscreteScheduler
from .guide import GuideBase, SimpleGuide

VAE_SCALE = 0.18215


@contextlib.contextmanager
def _gc_paused():
    was = gc.isenabled()
    if was:
        gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class StableDiffusionPipelineOutput():
    def __init__(self, images, nsfw_content_detected):
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected

    def __getitem__(self, key):
        if key in ('sample', 'images', 0):
            return self.images
        if key in ('nsfw_content_detected', 1):
            return self.nsfw_content_detected
        raise KeyError(key)


class FlexPipeline():
    def __init__(self, vae, clip, tokenizer, unet, scheduler):
        scheduler = scheduler.set_format('pt')
        # pipeline/flex.py:57-70: a scheduler config that HAS a steps_offset other than 1 is rewritten
        # to 1 (with a DeprecationWarning) through the scheduler's `_internal_dict`
        cfg = scheduler.config
        if hasattr(cfg, 'steps_offset') and cfg['steps_offset'] != 1:
            warnings.warn(f'The configuration file of this scheduler: {scheduler} is outdated. '
                          f'`steps_offset` should be set to 1 instead of {cfg["steps_offset"]}.',
                          DeprecationWarning)
            fixed = dict(cfg)
            fixed['steps_offset'] = 1
            setattr(scheduler, '_internal_dict', type(cfg)(fixed))
        self.vae = vae
        self.clip = clip
        self.tokenizer = tokenizer
        self.unet = unet
        self.scheduler = scheduler
        self.device = getattr(unet, 'device', torch.device('cuda'))
        self.last_latents: Optional[torch.Tensor] = None
        self.last_images: Optional[torch.Tensor] = None
        # default: replay the UNet forward of the fused loop from a captured HIP graph -- one host call per step, so a
        # host stall cannot starve the device; 0.3-0.6 % faster per forward than the launch plan on the 311-launch
        # forward (profiles/r04_session_ab.txt sec. 2).  If capture fails, `_unet_eps` switches this pipeline to the
        # launch plan (`graph_fallback` says why).
        self.use_graph = True
        self.graph_fallback: Optional[str] = None
        self._graphs = {}
        self._lat_bufs = {}
        # the launch plan (hip.Plan): identical kernels / order / results to the eager front, ~1/5 of its host time;
        # used when use_graph is off or fell back
        self.use_plan = True
        self._plans = {}
        # (timestep -> row, [steps][sum Cout] table) of the ResBlocks' time-embedding biases for the running request's timesteps
        self._temb_tab = None
        # opt-in (bench.py, Runner(pause_gc=True)): keep the cyclic GC off across the denoising loop.
        # Off by default: a drop-in must not change interpreter-global state of someone else's process.
        self.pause_gc = False

    @classmethod
    def from_pretrained(cls, sd_dir, clip_dir=None, tokenizer_dir=None, preset: str = 'sd15',
                        device='cuda', scheduler=None, text_cleanup: str = 'basic', **_):
        '''Local files only (there is no hub access): `sd_dir` is a diffusers-layout checkpoint
        directory (unet/, vae/, optionally tokenizer/), `clip_dir` a CLIPModel directory -- what the
        reference's `Runner.__init__` obtains from the hub (utils.py:59-71).  See
        `flexdiffuse_amd.build.from_directories`.'''
        from .. import build
        return build.from_directories(sd_dir, clip_dir, tokenizer_dir, preset=preset, device=device,
                                      scheduler=scheduler, text_cleanup=text_cleanup)[0]

    def to(self, device):
        self.device = torch.device(device)
        return self

    def progress_bar(self, iterable):
        return iterable

    @staticmethod
    def numpy_to_pil(images: np.ndarray):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype('uint8')
        return [Image.fromarray(image) for image in images]

    def enable_attention_slicing(self, slice_size: Optional[Union[str, int]] = 'auto'):
        if slice_size == 'auto':
            slice_size = self.unet.config['attention_head_dim'] // 2
        self.unet.set_attention_slice(slice_size)

    def disable_attention_slicing(self):
        self.enable_attention_slicing(None)

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        '''latents -> device image tensor (B,3,H,W) fp32 in [0,1] (pipeline/flex.py:117-120).'''
        img = self.vae.decode_nhwc(latents, scale=1.0 / VAE_SCALE)
        return ops.nhwc_to_nchw(img.t, img.B, 3, img.H, img.W, 0.5, 0.5, True)

    def _decode_generic(self, latents: torch.Tensor) -> torch.Tensor:
        '''pipeline/flex.py:116-120 for any object with the diffusers surface `decode(z).sample`
        (NCHW): scale, decode, (x/2 + 0.5).clamp(0, 1) -- the affine + clamp on the library's layout
        kernel with the NCHW tensor read as B*C one-channel maps.'''
        z = ops.axpby(latents.to(self.device, torch.float32), None, 1.0 / VAE_SCALE, 0.0)
        image = self.vae.decode(z).sample.to(torch.float32).contiguous()
        B, C, H, W = image.shape
        return ops.nhwc_to_nchw(image.view(-1, 1), B * C, 1, H, W, 0.5, 0.5, True).view(B, C, H, W)

    def _latents_to_image(self, latents: torch.Tensor, pil: bool = True):
        image = self.decode_latents(latents) if hasattr(self.vae, 'decode_nhwc') \
            else self._decode_generic(latents)
        self.last_images = image
        image = image.cpu().permute(0, 2, 3, 1).numpy()
        if pil:
            return self.numpy_to_pil(image)
        return image

    def _temb_row(self, t) -> torch.Tensor:
        '''[1][sum Cout] fp32 time-embedding biases of timestep t: a row of the table the running request computed for all of its
        timesteps at once (`__call__`), or computed here for a timestep outside it.'''
        tab = self._temb_tab
        if tab is not None:
            i = tab[0].get(float(t))
            if i is not None:
                return tab[1][i:i + 1]
        return self.unet.time_bias(float(t), 1)

    def _unet_eps(self, latents: torch.Tensor, t: int, ctx: torch.Tensor, rep: int) -> torch.Tensor:
        '''UNet noise prediction (NHWC fp32) for the fused loop; graph-replayed when enabled.
        `latents` must be the loop's persistent buffer (updated in place by the DDIM kernel).  The time embedding does not
        depend on the latents: the replayed forward reads the ResBlocks' time biases from a persistent buffer that is refreshed
        per step from the request's table (`_temb_row`) -- the three time-embedding GEMMs are not part of the step.'''
        # (per-launch event timing -- bench.py's roofline leg -- cannot see inside a graph: it replays the launch plan,
        # the same kernels in the same order issued by one host call)
        if (self.use_plan and not self.use_graph) or (self.use_graph and hip.prof_is_on()):
            return self._unet_eps_plan(latents, t, ctx, rep)
        Be = latents.shape[0] * rep
        if not self.use_graph:
            return self.unet.forward_nhwc(latents, t, ctx, rep=rep, temb=self._temb_row(t).expand(Be, -1).contiguous())
        self.unet.set_context(ctx)               # eager, in place: the graph reads these buffers
        # ctx_generation changes when the UNet had to reallocate its cached K / V^T (a call with
        # another context shape in between): a graph captured before that reads freed memory
        key = (latents.data_ptr(), tuple(latents.shape), rep, tuple(ctx.shape),
               getattr(self.unet, 'ctx_generation', 0))
        entry = self._graphs.get(key)
        row = self._temb_row(t)
        if entry is None:
            self._graphs = {}                    # drop the previous graph (its private pool holds GBs) before capturing another
            temb_static = torch.empty((Be, row.shape[1]), dtype=torch.float32, device=latents.device)
            # the first call runs eagerly (lazy one-time setup inside the kernels' launchers),
            # the second is captured
            temb_static.copy_(row.expand_as(temb_static))
            self.unet.forward_nhwc(latents, t, ctx, rep=rep, temb=temb_static)
            torch.cuda.synchronize()
            try:
                graph = torch.cuda.CUDAGraph()
                # thread_local: only THIS thread's calls are checked against the capture -- a process-group watchdog
                # thread (RCCL, N > 1) polling its events must not invalidate it
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    eps = self.unet.forward_nhwc(latents, t, ctx, rep=rep, temb=temb_static)
            except Exception as ex:      # noqa: BLE001 -- capture is an optimisation: never fail the request over it
                self.graph_fallback = f'HIP-graph capture failed ({type(ex).__name__}: {str(ex)[:200]}); running on the launch plan'
                warnings.warn(self.graph_fallback, RuntimeWarning)
                self.use_graph, self.use_plan, self._graphs = False, True, {}
                try:                     # a stream left in (or invalidated by) the failed capture must not fail the request either
                    if not torch.cuda.is_current_stream_capturing():
                        torch.cuda.synchronize()
                except Exception:        # noqa: BLE001
                    pass
                return self._unet_eps_plan(latents, t, ctx, rep)
            entry = (graph, temb_static, eps, ctx)
            self._graphs = {key: entry}          # keep one graph (its pool holds GBs)
        graph, temb_static, eps, ctx_ref = entry
        temb_static.copy_(row.expand_as(temb_static))
        graph.replay()
        return eps

    def _unet_eps_plan(self, latents: torch.Tensor, t: int, ctx: torch.Tensor, rep: int) -> torch.Tensor:
        '''The UNet forward through its launch plan.  `latents` is the loop's persistent buffer; the ResBlocks' time biases
        live in a persistent buffer refreshed before every replay (`_temb_row`); the context's
        K / V^T are (re)projected eagerly, in place, before the replay.  The recording run executes
        inside a private torch memory pool that the plan entry keeps alive, so every intermediate
        address the recorded launches use stays reserved for them.'''
        self.unet.set_context(ctx)
        key = (latents.data_ptr(), tuple(latents.shape), rep, tuple(ctx.shape),
               getattr(self.unet, 'ctx_generation', 0), hip.stream().value)
        entry = self._plans.get(key)
        row = self._temb_row(t)
        if entry is None:
            self._plans = {}                     # one plan at a time (its pool holds the activations)
            temb_dev = torch.empty((latents.shape[0] * rep, row.shape[1]), dtype=torch.float32, device=latents.device)
            temb_dev.copy_(row.expand_as(temb_dev))
            # first call eager: one-time setup inside the launchers, scratch buffers of ops.py
            self.unet.forward_nhwc(latents, t, ctx, rep=rep, temb=temb_dev)
            if self.unet.ctx_generation != key[4]:
                key = key[:4] + (self.unet.ctx_generation,) + key[5:]
            pool = torch.cuda.MemPool()
            plan = hip.Plan()
            with torch.cuda.use_mem_pool(pool, device=latents.device), plan.record():
                eps = self.unet.forward_nhwc(latents, t, ctx, rep=rep, temb=temb_dev)
            self._plans = {key: (plan, temb_dev, eps, pool, len(plan))}
            return eps                           # the recording run also executed
        plan, temb_dev, eps = entry[:3]
        temb_dev.copy_(row.expand_as(temb_dev))
        plan.replay()
        return eps

    def plan_launches(self):
        '''Recorded launches of the current plan (None before the first fused step).'''
        for e in self._plans.values():
            return e[4]
        return None

    def loop_latents(self, latents: torch.Tensor) -> torch.Tensor:
        '''The persistent per-shape latent buffer of the fused loop, filled with `latents` (a captured
        graph / recorded plan reads this address every step).'''
        latents = latents.to(self.device, torch.float32)
        buf = self._lat_bufs.get(tuple(latents.shape))
        if buf is None:
            buf = torch.empty_like(latents)
            self._lat_bufs = {tuple(latents.shape): buf}
        buf.copy_(latents)
        return buf

    def _randn(self, shape, generator):
        gdev = getattr(generator, 'device', torch.device('cpu')) if generator is not None \
            else torch.device('cpu')
        return torch.randn(shape, generator=generator, device=gdev,
                           dtype=torch.float32).to(self.device)

    @torch.no_grad()
    def __call__(self,
                 guide: GuideBase,
                 init_image=None,
                 init_size: Tuple[int, int] = (512, 512),
                 strength: float = 0.6,
                 eta: float = 0.0,
                 generator: Optional[torch.Generator] = None,
                 output_type: str = 'pil',
                 return_dict: bool = True,
                 debug: bool = False,
                 latents: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None):
        '''Arguments and defaults of pipeline/flex.py:127-137, plus two optional tensors for sharded
        runs (flexdiffuse_amd.dist): `latents` -- txt2img initial latents (B,4,h,w) instead of the
        pipeline's own randn; `noise` -- img2img add_noise rows (B,4,h,w) instead of its own randn
        (the rank's rows of the global batch's draw, `dist.global_img2img_noise`).'''
        if strength < 0 or strength > 1:
            raise ValueError(
                f'The value of strength should in [0.0, 1.0] but is {strength}')
        batch_size = guide.batch_size
        self.scheduler.set_timesteps(guide.steps)
        assert self.scheduler.timesteps is not None

        if init_image is not None:
            if not isinstance(init_image, torch.Tensor):
                init_image = preprocess(init_image)
            init_image = init_image.to(self.device)
            dist = self.vae.encode(init_image).latent_dist
            init_latents = dist.sample(generator=generator)
            init_latents = ops.axpby(init_latents, None, VAE_SCALE, 0.0)
            init_latents = torch.cat([init_latents] * batch_size)
            offset = self.scheduler.config.get('steps_offset', 0)
            init_timestep = int(guide.steps * strength) + offset
            init_timestep = min(init_timestep, guide.steps)
            if isinstance(self.scheduler, LMSDiscreteScheduler):
                t_noise = guide.steps - init_timestep     # pipeline/flex.py:200-204: an index
            else:
                t_noise = int(self.scheduler.timesteps[-init_timestep])
            # one level per sample, as a long tensor (pipeline/flex.py:201-209); built on the host
            t_noise = torch.tensor([t_noise] * batch_size, dtype=torch.long)
            noise = self._randn(init_latents.shape, generator) if noise is None \
                else noise.to(self.device, torch.float32)
            if tuple(noise.shape) != tuple(init_latents.shape):
                raise ValueError(f'noise {tuple(noise.shape)} does not match the latents {tuple(init_latents.shape)}')
            init_latents = self.scheduler.add_noise(init_latents, noise, t_noise)
            t_start = max(guide.steps - init_timestep + offset, 0)
        else:
            height, width = init_size
            channels = self.unet.in_channels
            shape = (batch_size, channels, height // 8, width // 8)
            init_latents = self._randn(shape, generator) if latents is None \
                else latents.to(self.device, torch.float32).clone()
            self.scheduler.set_timesteps(guide.steps)
            if isinstance(self.scheduler, LMSDiscreteScheduler):   # pipeline/flex.py:236-238
                init_latents = ops.axpby(init_latents, None, float(self.scheduler.sigmas[0]), 0.0)
            t_start = 0

        accepts_eta = 'eta' in set(inspect.signature(self.scheduler.step).parameters.keys())
        extra_step_kwargs = {'eta': eta} if accepts_eta else {}

        latents = init_latents.contiguous()
        all_latents = [init_latents] if debug else None
        fused = (type(guide).noise_pred is SimpleGuide.noise_pred
                 and isinstance(self.scheduler, DDIMScheduler) and not eta
                 and hasattr(self.unet, 'forward_nhwc'))
        B, C, H, W = latents.shape
        # SimpleGuide with ANY other scheduler (PNDM -- what the reference's Runner passes, utils.py:70 -- LMS, DDIM with
        # eta): the UNet forward still comes from the launch plan / graph; only the scheduler arithmetic stays generic
        planned = (not fused and type(guide).noise_pred is SimpleGuide.noise_pred and hasattr(self.unet, 'forward_nhwc')
                   and (self.use_graph or self.use_plan) and not debug)
        if fused and (self.use_graph or self.use_plan) and not debug:
            # persistent latent buffer: the captured UNet graph / recorded plan reads this address
            latents = self.loop_latents(latents)
        is_lms = isinstance(self.scheduler, LMSDiscreteScheduler)
        self._temb_tab = None
        if (fused or planned) and hasattr(self.unet, 'time_bias_table'):
            # the time embedding depends on t only: all of the request's timesteps in one pass (three GEMMs over len(timesteps) rows)
            # instead of three GEMMs per step
            ts = [float(t) for t in self.scheduler.timesteps[t_start:]]
            keys = {}
            for t in ts:
                keys.setdefault(t, len(keys))
            self._temb_tab = (keys, self.unet.time_bias_table(list(keys)))
        # The host only has to stay ahead of the device queue.  A generation-2 collection of the
        # cyclic GC walks every tracked object of the process (~175 k with the SD1.5 weights:
        # ~40 ms) and drains that queue, so collections wait until the images are decoded.
        with (_gc_paused() if self.pause_gc else contextlib.nullcontext()):
            for i, t in enumerate(self.progress_bar(self.scheduler.timesteps[t_start:])):
                if fused:
                    cfg = guide.classifier_free_guidance
                    if debug:
                        eps = self.unet.forward_nhwc(latents, int(t), guide.stacked_embeds(),
                                                     rep=2 if cfg else 1)
                    else:
                        eps = self._unet_eps(latents, int(t), guide.stacked_embeds(), 2 if cfg else 1)
                    coef = self.scheduler.step_coefficients(int(t))[:4]
                    if debug:
                        latents = latents.clone()
                    ops.cfg_ddim_step(latents, eps, B, C, H * W, cfg, guide.guidance, coef,
                                      self.scheduler.config['prediction_type'] == 'v_prediction')
                else:
                    t_index, model_input = t, latents
                    if is_lms:        # pipeline/flex.py:270-274: continuous-ODE input scaling
                        t_index = t_start + i
                        sigma = float(self.scheduler.sigmas[t_index])
                        model_input = ops.axpby(latents, None, 1.0 / ((sigma ** 2 + 1) ** 0.5), 0.0)
                    if planned:
                        cfg = guide.classifier_free_guidance
                        eps = self._unet_eps(self.loop_latents(model_input), float(t), guide.stacked_embeds(),
                                             2 if cfg else 1)
                        noise_pred = torch.empty((B, C, H, W), dtype=torch.float32, device=latents.device)
                        ops.cfg_ddim_step(None, eps, B, C, H * W, cfg, guide.guidance, do_step=False, eps_out=noise_pred)
                    else:
                        noise_pred = guide.noise_pred(model_input, t)
                    latents = self.scheduler.step(noise_pred, t_index, latents,
                                                  **extra_step_kwargs).prev_sample
                if all_latents is not None:
                    all_latents.append(latents)
            # the fused loop ran on the persistent per-shape buffer, which the next call overwrites:
            # hand out a copy, so a caller holding `pipe.last_latents` keeps the values it read
            persistent = any(latents is b for b in self._lat_bufs.values())
            self.last_latents = latents.clone() if persistent else latents

            if all_latents:
                batches = [self._latents_to_image(l, output_type == 'pil') for l in all_latents]
                if isinstance(batches[0], list):
                    batch_images = [im for ib in batches for im in ib]
                else:
                    batch_images = np.concatenate(batches, axis=0)
            else:
                batch_images = self._latents_to_image(latents, output_type == 'pil')

        if not return_dict:
            return (batch_images, False)
        return StableDiffusionPipelineOutput(images=batch_images,
                                             nsfw_content_detected=[False for _ in batch_images])
