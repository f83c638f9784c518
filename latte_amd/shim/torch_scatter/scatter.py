import sys

from latte_amd.segment import scatter  # noqa: F401


class _Callable(sys.modules[__name__].__class__):
    # the reference does `import torch_scatter.scatter as ts_scatter` and then calls ts_scatter(...):
    # with the real wheel that name resolves to the function; make this module callable the same way.
    def __call__(self, *a, **k):
        return scatter(*a, **k)


sys.modules[__name__].__class__ = _Callable
