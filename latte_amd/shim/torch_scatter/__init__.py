"""Drop-in for `torch_scatter.scatter` as Latte uses it (`import torch_scatter.scatter as ts_scatter`,
latte/models/hr_module.py:5,236,237,252): segment sum / mean / max over dim 0, rows = index.max()+1."""
from .scatter import scatter  # noqa: F401
