'''Recording stand-ins for the third-party objects `FlexPipeline.__call__` drives -- TEST
INFRASTRUCTURE ONLY.

The reference's pipeline/flex.py:126-310 is pure control flow around five duck-typed objects
(guide, scheduler, vae, unet, generator).  `tests/golden/make_flexcall_goldens.py` runs the
reference's OWN `FlexPipeline.__call__` with the stubs below in place of diffusers' classes and
stores the call trace (which method, in which order, with which tensors / timesteps / indices)
and the returned images.  The tests replay the same stubs through
  * `oracle/flexcall_ref.py`                 (CPU, `-m "not gpu"`)
  * `flexdiffuse_amd.pipeline.flex.FlexPipeline` (device, `-m gpu`)
and require the same trace.  The stubs' arithmetic is arbitrary but value-sensitive: every
output depends on the timestep / index it was called with, so a wrong `t_start`, `t_index`,
`add_noise` level or draw order changes the numbers downstream, not just the log.

Nothing here restates reference code: the stubs are this repo's own fixtures.
'''
import types

import numpy as np
import torch
import torch.nn.functional as F


BIG = 2048        # arrays above this many elements are stored as a digest (fixtures stay small)


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def digest(a):
    '''Small arrays as they are; big ones as {shape, ~512 strided elements, sum, sum of squares}.
    Returns a dict name-suffix -> ndarray.'''
    a = _np(a)
    if a.size <= BIG:
        return {'': a}
    flat = a.reshape(-1).astype(np.float64)
    return {'__shape': np.array(a.shape), '__sub': flat[::max(1, a.size // 512)].astype(np.float32),
            '__sum': np.array(flat.sum()), '__sumsq': np.array((flat * flat).sum())}


class Trace():
    '''Ordered list of (kind, {name: ndarray}) events.'''

    def __init__(self):
        self.events = []

    def add(self, kind, **arrays):
        ev = {}
        for k, v in arrays.items():
            for suffix, a in digest(v).items():
                ev[k + suffix] = a
        self.events.append((kind, ev))

    def kinds(self):
        return [k for k, _ in self.events]

    def to_npz(self, prefix, out):
        out[f'{prefix}/kinds'] = np.array(self.kinds() or [''])
        for i, (_, arrs) in enumerate(self.events):
            for name, a in arrs.items():
                out[f'{prefix}/e{i:03d}/{name}'] = a

    @staticmethod
    def from_npz(prefix, z):
        tr = Trace()
        kinds = [str(k) for k in z[f'{prefix}/kinds'] if str(k)]
        for i, kind in enumerate(kinds):
            head = f'{prefix}/e{i:03d}/'
            tr.events.append((kind, {k[len(head):]: z[k] for k in z.files if k.startswith(head)}))
        return tr


class FrozenConfig(dict):
    '''dict whose keys also read as attributes (what `hasattr(scheduler.config, 'steps_offset')`
    at pipeline/flex.py:57 relies on).'''

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def _cpu_randn(shape, generator):
    dev = getattr(generator, 'device', torch.device('cpu')) if generator is not None else 'cpu'
    return torch.randn(shape, generator=generator, device=dev, dtype=torch.float32)


class StubDist():
    def __init__(self, mean, trace):
        self.mean, self.trace = mean, trace

    def sample(self, generator=None):
        noise = _cpu_randn(tuple(self.mean.shape), generator).to(self.mean.device)
        self.trace.add('latent_dist.sample', noise=noise)
        return self.mean + 0.1 * noise


class StubVAE():
    '''encode: 8x average pool of the image -> 4 channels; decode: nearest 8x of 3 channels.'''

    def __init__(self, trace):
        self.trace = trace

    def encode(self, x):
        self.trace.add('vae.encode', x=x)
        m = F.avg_pool2d(x.float(), 8)
        mean = torch.cat([m, m[:, :1] * -0.5], dim=1)
        return types.SimpleNamespace(latent_dist=StubDist(mean, self.trace))

    def decode(self, z):
        self.trace.add('vae.decode', z=z)
        img = F.interpolate(z[:, :3].float(), scale_factor=8, mode='nearest') * 0.35 + 0.05 * z[:, 3:4].mean()
        return types.SimpleNamespace(sample=img)


class StubGuide():
    '''Only what the pipeline reads: batch_size, steps, noise_pred (pipeline/flex.py:174,177,277).'''

    def __init__(self, trace, batch_size, steps, seed):
        self.trace, self.batch_size, self.steps = trace, batch_size, steps
        self.g = torch.Generator().manual_seed(seed)

    def noise_pred(self, latents, step):
        self.trace.add('noise_pred', latents=latents, t=_scalar(step))
        r = torch.randn(tuple(latents.shape), generator=self.g).to(latents.device)
        return 0.5 * r + 0.1 * latents * (1.0 + 1e-3 * float(_scalar(step)))


def _scalar(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().reshape(-1)[0].item() if v.numel() == 1 else v.detach().cpu().numpy()
    return v.item() if hasattr(v, 'item') else v


def make_scheduler(base, kind, trace, steps_offset=None, numpy_timesteps=False):
    '''A stub scheduler class derived from `base` (the class the pipeline under test does its
    isinstance checks against) -- kind 'ddim' (step takes eta), 'pndm' (no eta), 'lms' (sigmas,
    index-based step / add_noise).  `steps_offset=None`: the config has no such key (diffusers
    0.3.0).'''

    class _Stub(base):
        def __init__(self):                       # never runs the base constructor
            cfg = {'num_train_timesteps': 1000}
            if steps_offset is not None:
                cfg['steps_offset'] = steps_offset
            self._internal_dict = FrozenConfig(cfg)
            self.trace = trace
            self.kind = kind
            self.timesteps = None
            self.sigmas = None

        @property
        def config(self):
            return self._internal_dict

        def set_format(self, tensor_format='pt'):
            self.trace.add('set_format', fmt=np.array(str(tensor_format)))
            return self

        def set_timesteps(self, n):
            off = self.config.get('steps_offset', 0)
            self.trace.add('set_timesteps', n=n, offset_in_config=off)
            if kind == 'lms':
                ts = np.linspace(999, 0, n, dtype=np.float32)
                self.sigmas = torch.from_numpy(
                    np.concatenate([np.linspace(14.6, 0.03, n), [0.0]]).astype(np.float32))
                self.timesteps = ts if numpy_timesteps else torch.from_numpy(ts.copy())
            else:
                ts = (np.arange(0, 1000, 1000 // n)[::-1].copy() + off).astype(np.int64)
                self.timesteps = ts if numpy_timesteps else torch.from_numpy(ts)

        def add_noise(self, original, noise, timesteps):
            self.trace.add('add_noise', original=original, noise=noise, timesteps=timesteps)
            tt = torch.as_tensor(_np(timesteps), dtype=torch.float32).reshape(-1, 1, 1, 1).to(original.device)
            return original * (1.0 - tt / 2000.0) + noise.to(original.device) * (0.05 + tt / 1000.0)

        def _step(self, model_output, timestep, sample, eta):
            self.trace.add('step', model_output=model_output, timestep=_scalar(timestep), sample=sample,
                           eta=np.array(-1.0 if eta is None else float(eta)))
            t = float(_scalar(timestep))
            prev = sample * (1.0 - 1e-4 * t) - model_output * (0.05 + 1e-5 * t)
            if eta:
                prev = prev + 0.01 * float(eta)
            return types.SimpleNamespace(prev_sample=prev)

    if kind == 'ddim':
        def step(self, model_output, timestep, sample, eta: float = 0.0):
            return self._step(model_output, timestep, sample, eta)
    else:
        def step(self, model_output, timestep, sample):
            return self._step(model_output, timestep, sample, None)
    _Stub.step = step
    _Stub.__name__ = f'Stub{kind.upper()}Scheduler'
    return _Stub()


def guide_image(w, h, seed):
    '''Seeded low-passed uint8 RGB PIL image.'''
    from PIL import Image
    rng = np.random.default_rng(seed)
    a = rng.uniform(0, 255, (h // 4 + 1, w // 4 + 1, 3)).astype(np.float32)
    a = np.kron(a, np.ones((4, 4, 1), np.float32))[:h, :w]
    return Image.fromarray(a.astype(np.uint8))


# name: dict(kind, steps_offset, B, steps, call kwargs..., init=(w, h) of a PIL init image or None)
CASES = {
    'txt2img_ddim': dict(kind='ddim', steps_offset=None, B=2, steps=6, init=None,
                         kw=dict(init_size=(64, 96), output_type='np')),
    'txt2img_ddim_eta_tuple_pil': dict(kind='ddim', steps_offset=None, B=1, steps=5, init=None,
                                       kw=dict(init_size=(64, 64), eta=0.3, return_dict=False)),
    'img2img_ddim_nooffset': dict(kind='ddim', steps_offset=None, B=2, steps=10, init=(100, 80),
                                  kw=dict(strength=0.6, output_type='np')),
    'img2img_ddim_offset1': dict(kind='ddim', steps_offset=1, B=2, steps=10, init=(100, 80),
                                 kw=dict(strength=0.6, output_type='np')),
    'img2img_ddim_offset0_rewritten': dict(kind='ddim', steps_offset=0, B=1, steps=10, init=(96, 96),
                                           kw=dict(strength=0.45, output_type='np')),
    'img2img_strength0': dict(kind='ddim', steps_offset=None, B=1, steps=8, init=(96, 96),
                              kw=dict(strength=0.0, output_type='np')),
    'img2img_strength1_offset1': dict(kind='ddim', steps_offset=1, B=1, steps=8, init=(96, 96),
                                      kw=dict(strength=1.0, output_type='np')),
    'txt2img_lms': dict(kind='lms', steps_offset=None, B=2, steps=6, init=None,
                        kw=dict(init_size=(64, 64), output_type='np')),
    'img2img_lms': dict(kind='lms', steps_offset=None, B=2, steps=10, init=(80, 100),
                        kw=dict(strength=0.5, output_type='np')),
    'txt2img_pndm_eta_not_forwarded': dict(kind='pndm', steps_offset=1, B=1, steps=5, init=None,
                                           kw=dict(init_size=(64, 64), eta=0.5, output_type='np')),
    'txt2img_debug_np': dict(kind='ddim', steps_offset=None, B=2, steps=3, init=None,
                             kw=dict(init_size=(64, 64), output_type='np', debug=True)),
    'img2img_debug_pil': dict(kind='pndm', steps_offset=None, B=1, steps=4, init=(64, 64),
                              kw=dict(strength=0.8, debug=True)),
}


def run_case(name, pipeline_factory, sched_bases, device='cpu'):
    '''Builds the stubs of case `name`, constructs the pipeline through
    `pipeline_factory(vae, clip, tokenizer, unet, scheduler)`, calls it, and returns
    (trace, images as ndarray, nsfw flags or None, warnings raised by the constructor).'''
    import warnings
    c = CASES[name]
    trace = Trace()
    sched = make_scheduler(sched_bases[c['kind']], c['kind'], trace, c['steps_offset'])
    vae = StubVAE(trace)
    unet = types.SimpleNamespace(in_channels=4, config={'attention_head_dim': 8}, device=torch.device(device))
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        pipe = pipeline_factory(vae, object(), object(), unet, sched)
    ctor_warnings = [w.category.__name__ for w in wlist]
    trace.add('ctor', steps_offset_after=np.array(sched.config.get('steps_offset', -1)))
    guide = StubGuide(trace, c['B'], c['steps'], seed=1234)
    gen = torch.Generator('cpu').manual_seed(4321)
    kw = dict(c['kw'])
    if c['init'] is not None:
        kw['init_image'] = guide_image(*c['init'], seed=5)
    res = pipe(guide=guide, generator=gen, **kw)
    if isinstance(res, tuple):
        images, flags = res
        trace.add('returned_tuple', second=np.array(bool(flags)))
        flags = None
    else:
        images, flags = res.images, list(res.nsfw_content_detected)
    if isinstance(images, list):
        images = np.stack([np.asarray(im) for im in images])
    return trace, np.asarray(images), flags, ctor_warnings


def assert_same(got, want, what, rtol=2e-6, atol=2e-6):
    '''got / want: dicts name -> ndarray as produced by `digest` (or Trace events).'''
    assert sorted(got) == sorted(want), f'{what}: fields {sorted(got)} != {sorted(want)}'
    for k in want:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        assert g.shape == w.shape, f'{what}.{k}: shape {g.shape} != {w.shape}'
        if w.dtype.kind in 'iub' or k.endswith('__shape'):
            assert np.array_equal(g, w), f'{what}.{k}: {g} != {w}'
        elif w.dtype.kind in 'US':
            assert str(g) == str(w), f'{what}.{k}: {g} != {w}'
        elif k.endswith(('__sum', '__sumsq')):
            assert abs(float(g) - float(w)) <= 1e-5 * max(1.0, abs(float(w))), f'{what}.{k}: {g} != {w}'
        else:
            assert g.dtype.kind == w.dtype.kind, f'{what}.{k}: dtype {g.dtype} != {w.dtype}'
            np.testing.assert_allclose(g, w, rtol=rtol, atol=atol, err_msg=f'{what}.{k}')


def assert_same_trace(got: Trace, want: Trace, name):
    assert got.kinds() == want.kinds(), f'{name}: call order\n got  {got.kinds()}\n want {want.kinds()}'
    for i, ((kind, g), (_, w)) in enumerate(zip(got.events, want.events)):
        assert_same(g, w, f'{name}[{i}:{kind}]')
