from . import collate, quantize  # noqa: F401
