from latte_amd.tensor import cat  # noqa: F401
