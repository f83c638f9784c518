"""Drop-in for the slice of `torchsparse` (mit-han-lab/torchsparse v1.4.0, reference install.sh:9) that Latte's
3-D hot path imports.  Every op runs in liblatte_hip.so (hand-written HIP for gfx950); nothing here is a port of
torchsparse's CUDA sources."""
from .operators import cat
from .tensor import PointTensor, SparseTensor
from . import nn, utils  # noqa: F401

__version__ = "1.4.0+latte_amd"
__all__ = ["SparseTensor", "PointTensor", "cat", "nn", "utils"]
