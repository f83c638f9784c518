"""Generates tests/golden/spvcnn_state_keys.json: parameter / buffer names and shapes of the reference's OWN Net3DSeg +
SPVCNN module tree (latte/models/xmuda_arch.py:289-342, latte/models/spvcnn.py:16-114, latte/models/basic_blocks.py),
instantiated with `torchsparse` resolving to latte_amd/shim -- i.e. the Level-1 drop-in: the reference's model files run
unmodified on the shim's spnn.Conv3d / BatchNorm / ReLU.  Names + shapes only; no reference source text, no weights.

Runs ONLY in the build container (needs /root/reference, CPU is enough: constructing modules launches no kernel).
usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_state_keys.py
"""
import importlib
import json
import os
import sys
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import latte_amd  # noqa: E402

latte_amd.install_shims()                       # `import torchsparse` -> latte_amd/shim/torchsparse
for m in ["timm", "timm.models", "timm.models.layers", "IPython", "open3d", "attr", "matplotlib", "matplotlib.pyplot",
          "torchvision", "torchvision.models", "torchvision.models.resnet", "mmcv", "mmcv.cnn"]:
    try:
        importlib.import_module(m)
    except Exception:
        sys.modules[m] = MagicMock()

import torchsparse  # noqa: E402
assert "latte_amd" in torchsparse.__file__, torchsparse.__file__
from latte.models.spvcnn import SPVCNN  # noqa: E402  (the reference's file, unmodified)
from latte.models.xmuda_arch import Net3DSeg  # noqa: E402

out = {}
for cr, ncls in ((1.0, 10), (0.5, 9)):
    net = Net3DSeg(num_classes=ncls, dual_head=True, backbone_3d="SPVCNN", backbone_3d_kwargs={"cr": cr})
    assert isinstance(net.net_3d, SPVCNN)
    sd = net.state_dict()
    out[f"cr{cr}_C{ncls}"] = {"cr": cr, "num_classes": ncls, "keys": {k: list(v.shape) for k, v in sd.items()},
                              "modules": {n: type(m).__name__ for n, m in net.named_modules() if n}}
json.dump(out, open(os.path.join(HERE, "spvcnn_state_keys.json"), "w"), indent=0, sort_keys=True)
print({k: len(v["keys"]) for k, v in out.items()})
