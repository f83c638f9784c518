from latte_amd.quantize import sparse_quantize  # noqa: F401

__all__ = ["sparse_quantize"]
