from latte_amd.tensor import PointTensor, SparseTensor  # noqa: F401
