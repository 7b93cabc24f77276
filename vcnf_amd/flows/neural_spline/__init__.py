from .coupling import PiecewiseRationalQuadraticCoupling, PiecewiseRationalQuadraticCDF   # noqa: F401
from .wrapper import CoupledRationalQuadraticSpline                                       # noqa: F401
