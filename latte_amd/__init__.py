"""latte_amd -- MI355X-native implementation of Latte's 3-D sparse-voxel hot path.

Host-side mirror of the reference's Python module API; all sparse arithmetic runs in liblatte_hip.so
(hand-written HIP for gfx950) through the C-ABI in include/latte_hip.h."""
import os
import sys

__version__ = "0.1.0"
SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")


def install_shims() -> None:
    """Make `import torchsparse` / `import torch_scatter` resolve to the drop-in packages in latte_amd/shim, so the
    reference's latte/models/*.py and latte/tta/tools/latte.py run unmodified on the HIP path."""
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
