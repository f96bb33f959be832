'''Loader for the *reference* implementation -- THIS CONTAINER ONLY.

Used solely by the golden-vector generator scripts in this directory. It imports
`/root/reference/guidance.py` and `/root/reference/encode/clip.py` (public,
read-only) behind an in-memory torchvision shim so that their
outputs on seeded inputs can be written to `tests/golden/*.npz`. Nothing under
`tests/` run by pytest, `bench.py` or `__graft_entry__` imports this module, and
no reference source is copied into the repo.
'''
import sys
import types

import torch
import torch.nn.functional as F

REF = '/root/reference'


def load_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import transformers  # noqa: F401  (must come first: it find_spec()s torchvision)
    tvf = types.ModuleType('torchvision.transforms.functional')

    def center_crop(x, s):
        o = int(round((x.shape[-2] - s[0]) / 2.))
        p = int(round((x.shape[-1] - s[1]) / 2.))
        return x[..., o:o + s[0], p:p + s[1]]

    def resize(x, size, interpolation=None, antialias=None):
        return F.interpolate(x, size=size, mode='bicubic', align_corners=False,
                             antialias=bool(antialias))

    def normalize(x, m, s):
        return (x - torch.tensor(m).view(-1, 1, 1)) / torch.tensor(s).view(-1, 1, 1)

    tvf.center_crop = center_crop
    tvf.resize = resize
    tvf.normalize = normalize
    tvf.InterpolationMode = type('IM', (), {'BICUBIC': 'bicubic'})
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt,
                        'torchvision.transforms.functional': tvf})
    import guidance
    import encode.clip as eclip
    return guidance, eclip
