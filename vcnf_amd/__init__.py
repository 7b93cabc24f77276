"""vcnf_amd - MI355X-native coupling-flow transform engine.

Drop-in for the bijector hot path of telegraphroad/VCNF (normflow 1.2 fork):
the module tree, constructor signatures and state_dict keys of the reference's
coupling layers, with the arithmetic running in hand-written gfx950 HIP kernels
behind the C ABI of ``include/vcnf_hip.h``.  No CPU path: CPU tensors raise.

    import vcnf_amd as nf
    flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128) for _ in range(12)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).to("cuda")
    with torch.no_grad():
        log_q = model.log_prob(x)
"""
from ._lib import lib, lib_path, VcnfError, check_discriminant, check_saturation, range_redo_count   # noqa: F401
from . import utils, nets, flows, distributions                  # noqa: F401
from .core import NormalizingFlow, MultiscaleFlow                # noqa: F401
from .sharded import ShardedEvaluator, shard_bounds              # noqa: F401
from .graphs import GraphedFlow, GraphedTrainStep                # noqa: F401
from .fused import refresh_packed                                # noqa: F401

__version__ = "0.1.0"
