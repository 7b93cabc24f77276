from .coupling import PiecewiseRationalQuadraticCoupling, PiecewiseRationalQuadraticCDF   # noqa: F401
from .autoregressive import MaskedPiecewiseRationalQuadraticAutoregressive                # noqa: F401
from .wrapper import (CoupledRationalQuadraticSpline, CircularCoupledRationalQuadraticSpline,    # noqa: F401
                      AutoregressiveRationalQuadraticSpline, CircularAutoregressiveRationalQuadraticSpline)
