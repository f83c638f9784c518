from latte_amd.quantize import sparse_collate, sparse_collate_fn  # noqa: F401

__all__ = ["sparse_collate", "sparse_collate_fn"]
