'''flexdiffuse_amd -- MI355X-native (gfx950) implementation of flexdiffuse's image-guided
denoising hot path.  Exports the reference package's names (reference __init__.py:7-14).'''
from . import guidance as _guidance
from .encode import clip as _encode
from .pipeline import flex as _flex

CLIPEncoder = _encode.CLIPEncoder
GUIDE_ORDER_TEXT = _guidance.GUIDE_ORDER_TEXT
GUIDE_ORDER_ALIGN = _guidance.GUIDE_ORDER_ALIGN
GUIDE_ORDER_DIRECT = _guidance.GUIDE_ORDER_DIRECT
Guide = _guidance.Guide
Tweener = _guidance.Tweener
preprocess = _encode.preprocess
FlexPipeline = _flex.FlexPipeline

from .pipeline.guide import GuideBase, PromptGuide, SimpleGuide  # noqa: E402,F401
from .scheduler import DDIMScheduler, LMSDiscreteScheduler, PNDMScheduler  # noqa: E402,F401
from .tokenizer import SyntheticTokenizer  # noqa: E402,F401
from .build import build_models, load_state_dicts, load_tokenizer, synthetic_state_dicts  # noqa: E402,F401
from .utils import Runner, image_grid  # noqa: E402,F401
