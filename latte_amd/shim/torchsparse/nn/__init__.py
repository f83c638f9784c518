"""torchsparse.nn surface used by latte/models/basic_blocks.py:17-72 and latte/models/spvcnn.py:30-33.
Class names are kept (`BatchNorm`, `ReLU`, `Conv3d`): latte/common/utils/ema_util.py:23-33 selects the
trainable modules by class-name substring."""
from latte_amd.sparse_modules import BatchNorm, Conv3d, ReLU  # noqa: F401
from . import functional, utils  # noqa: F401

__all__ = ["Conv3d", "BatchNorm", "ReLU", "functional", "utils"]
