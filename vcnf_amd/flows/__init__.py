from .base import Flow, Reverse, Composite                                        # noqa: F401
from .reshape import Split, Merge, Squeeze                                        # noqa: F401
from .mixing import Permute, Invertible1x1Conv, LULinearPermute                                    # noqa: F401
from .normalization import ActNorm                                                # noqa: F401
from .affine import (AffineConstFlow, AffineCoupling, MaskedAffineFlow,           # noqa: F401
                     AffineCouplingBlock)
from .affine.glow import GlowBlock                                                # noqa: F401
from .affine.autoregressive import MaskedAffineAutoregressive                     # noqa: F401
from .neural_spline import (CoupledRationalQuadraticSpline,                       # noqa: F401
                            CircularCoupledRationalQuadraticSpline, AutoregressiveRationalQuadraticSpline,
                            CircularAutoregressiveRationalQuadraticSpline,
                            PiecewiseRationalQuadraticCoupling, PiecewiseRationalQuadraticCDF)
