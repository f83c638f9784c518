from latte_amd.functional import get_kernel_offsets  # noqa: F401

__all__ = ["get_kernel_offsets"]
