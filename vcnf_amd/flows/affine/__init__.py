from .coupling import AffineConstFlow, AffineCoupling, MaskedAffineFlow, AffineCouplingBlock   # noqa: F401
