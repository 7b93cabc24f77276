from . import base                                             # noqa: F401
from .base import BaseDistribution, DiagGaussian               # noqa: F401
