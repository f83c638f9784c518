"""Generates tests/golden/ema_util_returns.json: what the reference's OWN latte/common/utils/ema_util.py returns for
configure_model_norm / configure_model_all / create_ema_model (the nested [model, [lists]] its callers unpack,
latte/tta/tta.py:130-136, latte/tta/tools/latte.py:114-115) and which parameters end up trainable, on the reference's own
Net3DSeg + SPVCNN instantiated over the shim (as make_state_keys.py).  Names / flags only: no reference source text.

Runs ONLY in the build container (needs /root/reference; CPU is enough).
usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ema_util_golden.py
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

latte_amd.install_shims()
for m in ["timm", "timm.models", "timm.models.layers", "IPython", "open3d", "attr", "matplotlib", "matplotlib.pyplot",
          "torchvision", "torchvision.models", "torchvision.models.resnet", "mmcv", "mmcv.cnn"]:
    try:
        importlib.import_module(m)
    except Exception:
        sys.modules[m] = MagicMock()

import torch  # noqa: E402
from latte.common.utils import ema_util  # noqa: E402  (the reference's file, unmodified)
from latte.models.xmuda_arch import Net3DSeg  # noqa: E402


def net():
    torch.manual_seed(0)
    return Net3DSeg(num_classes=9, dual_head=True, backbone_3d="SPVCNN", backbone_3d_kwargs={"cr": 0.5})


def describe(model):
    return {"trainable": sorted(n for n, p in model.named_parameters() if p.requires_grad),
            "training": bool(model.training),
            "dropout_p": {n: float(m.p) for n, m in model.named_modules() if "dropout" in type(m).__name__.lower()},
            "bn_track": sorted(n for n, m in model.named_modules()
                               if "batchnorm" in type(m).__name__.lower() and m.track_running_stats)}


out = {}
for reset in (True, False):
    r = ema_util.configure_model_norm(net(), reset)
    out[f"configure_model_norm_{reset}"] = {"lists": r[1], "model": describe(r[0])}
    r = ema_util.configure_model_all(net(), reset)
    out[f"configure_model_all_{reset}"] = {"lists": r[1], "model": describe(r[0])}
student = ema_util.configure_model_norm(net(), True)[0]
r = ema_util.create_ema_model(student)
out["create_ema_model"] = {"lists": r[1], "model": describe(r[0]),
                           "requires_grad_any": any(p.requires_grad for p in r[0].parameters())}
# update_ema_norm / update_ema_all with and without the warm-up `iteration`
for fn in ("update_ema_norm", "update_ema_all"):
    for it in (None, 3):
        s, e = net(), net()
        with torch.no_grad():
            for p in s.parameters():
                p.add_(1.0)
        getattr(ema_util, fn)(e, s, 0.99, it)
        ref = net()
        delta = {n: float((pe - pr).abs().max()) for (n, pe), (_, pr) in zip(e.named_parameters(), ref.named_parameters())}
        out[f"{fn}_{it}"] = {"moved": {n: round(v, 6) for n, v in delta.items()}}
json.dump(out, open(os.path.join(HERE, "ema_util_returns.json"), "w"), indent=0, sort_keys=True)
print({k: (len(v.get("lists", [])), len(v.get("model", {}).get("trainable", []))) for k, v in out.items()})
