from .base import Flow, Reverse, Composite                                        # noqa: F401
from .reshape import Split, Merge, Squeeze                                        # noqa: F401
from .mixing import Permute                                                       # noqa: F401
from .affine import (AffineConstFlow, AffineCoupling, MaskedAffineFlow,           # noqa: F401
                     AffineCouplingBlock)
from .neural_spline import (CoupledRationalQuadraticSpline,                       # noqa: F401
                            PiecewiseRationalQuadraticCoupling, PiecewiseRationalQuadraticCDF)
