"""``archs`` surface of the reference (archs/__init__.py:1-19): the denoiser modules are
looked up by name -- ``globals()[arch['name']](arch)`` (trainer_SID.py:17) -- and
initialised with ``initialize_weights``."""
import torch.nn as nn

from .noise_flow import NoiseFlow  # noqa: F401
from .resunet import ResUnet  # noqa: F401
from .unet import UNetSeeInDark  # noqa: F401


def initialize_weights(net):
    """archs/__init__.py:12-19: N(0, 0.02) for Conv2d weight and bias and for
    ConvTranspose2d weight (its bias keeps the framework default)."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.normal_(0.0, 0.02)
        if isinstance(m, nn.ConvTranspose2d):
            m.weight.data.normal_(0.0, 0.02)
