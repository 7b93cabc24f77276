from .coupling import PiecewiseRationalQuadraticCoupling, PiecewiseRationalQuadraticCDF   # noqa: F401
from .wrapper import CoupledRationalQuadraticSpline, CircularCoupledRationalQuadraticSpline                                       # noqa: F401
