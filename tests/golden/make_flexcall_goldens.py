'''Golden call traces of the reference's OWN `FlexPipeline.__call__` -- runs in the BUILD
CONTAINER only.

`/root/reference/pipeline/flex.py` is imported behind an in-memory, NAME-ONLY `diffusers`
package (diffusers 0.3.0 is not installed): a `DiffusionPipeline` base that stores modules and
provides `device` / `progress_bar` / `numpy_to_pil` / `register_modules`, empty scheduler classes
for the pipeline's isinstance tests, a `FrozenDict`, and the output record.  None of it is
diffusers code; it only lets the reference file's import statements resolve.  The pipeline is
then called with the recording stubs of tests/flexcall_stubs.py for txt2img, img2img (with and
without `steps_offset`, incl. the constructor's 0 -> 1 rewrite, strength 0 and 1), the LMS
branches, eta forwarding, debug=True and return_dict=False.  What is stored is DATA: the ordered
call trace (method, tensors, timesteps / indices) and the returned images.

Usage:  python tests/golden/make_flexcall_goldens.py
'''
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from _ref_loader import load_reference  # noqa: E402
import flexcall_stubs as S  # noqa: E402


def install_diffusers_names():
    class FrozenDict(S.FrozenConfig):
        pass

    class DiffusionPipeline():
        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def device(self):
            return torch.device('cpu')

        def to(self, device):
            return self

        def progress_bar(self, it):
            return it

        @staticmethod
        def numpy_to_pil(images):
            from PIL import Image
            if images.ndim == 3:
                images = images[None, ...]
            images = (images * 255).round().astype('uint8')
            return [Image.fromarray(im) for im in images]

    class StableDiffusionPipelineOutput():
        def __init__(self, images, nsfw_content_detected):
            self.images, self.nsfw_content_detected = images, nsfw_content_detected

    mods = {n: types.ModuleType(n) for n in (
        'diffusers', 'diffusers.configuration_utils', 'diffusers.models', 'diffusers.pipeline_utils',
        'diffusers.schedulers', 'diffusers.pipelines', 'diffusers.pipelines.stable_diffusion')}
    mods['diffusers.configuration_utils'].FrozenDict = FrozenDict
    mods['diffusers.models'].AutoencoderKL = type('AutoencoderKL', (), {})
    mods['diffusers.models'].UNet2DConditionModel = type('UNet2DConditionModel', (), {})
    mods['diffusers.pipeline_utils'].DiffusionPipeline = DiffusionPipeline
    for n in ('DDIMScheduler', 'LMSDiscreteScheduler', 'PNDMScheduler'):
        setattr(mods['diffusers.schedulers'], n, type(n, (), {}))
    mods['diffusers.pipelines.stable_diffusion'].StableDiffusionPipelineOutput = StableDiffusionPipelineOutput
    sys.modules.update(mods)
    return mods


def main():
    load_reference()                      # torchvision shim + sys.path for /root/reference
    mods = install_diffusers_names()
    import pipeline.flex as rflex         # the reference's own file
    sch = mods['diffusers.schedulers']
    bases = {'ddim': sch.DDIMScheduler, 'pndm': sch.PNDMScheduler, 'lms': sch.LMSDiscreteScheduler}
    out = {'names': np.array(list(S.CASES))}
    for name in S.CASES:
        trace, images, flags, ctor_warnings = S.run_case(name, rflex.FlexPipeline, bases)
        trace.to_npz(f'{name}/trace', out)
        for suffix, a in S.digest(images).items():
            out[f'{name}/images{suffix}'] = a
        out[f'{name}/images_dtype'] = np.array(str(images.dtype))
        out[f'{name}/flags'] = np.array(flags if flags is not None else [], dtype=bool)
        out[f'{name}/ctor_warnings'] = np.array(ctor_warnings or [''])
        print(f'{name:36s} {len(trace.events):3d} events  images {images.shape} {images.dtype}  '
              f'warn {ctor_warnings}')
    # the ValueError of pipeline/flex.py:170-172
    for bad in (-0.1, 1.5):
        try:
            S.CASES['_bad'] = dict(kind='ddim', steps_offset=None, B=1, steps=4, init=None,
                                   kw=dict(strength=bad, init_size=(64, 64)))
            S.run_case('_bad', rflex.FlexPipeline, bases)
            raise SystemExit('expected ValueError')
        except ValueError as ex:
            out[f'valueerror/{bad}'] = np.array(str(ex))
        finally:
            S.CASES.pop('_bad', None)
    path = os.path.join(HERE, 'flexcall_goldens.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
