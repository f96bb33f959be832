from .guide import CompositeGuide  # noqa: F401
from .schema import EntitySchema, Schema  # noqa: F401
