from .flex import FlexPipeline  # noqa: F401
from .guide import GuideBase, PromptGuide, SimpleGuide  # noqa: F401
