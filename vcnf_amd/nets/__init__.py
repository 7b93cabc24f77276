from .mlp import MLP                                   # noqa: F401
from .resnet import ResidualBlock, ResidualNet, ConvResidualNet         # noqa: F401
from .cnn import ConvNet2d                             # noqa: F401
from .made import MADE                                 # noqa: F401
