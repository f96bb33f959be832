from .clip import CLIPEncoder, preprocess  # noqa: F401
